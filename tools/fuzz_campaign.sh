cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_fuzz.txt; : > $O
run() { echo "== $*" >> $O; timeout 1500 python "$@" 2>&1 | tail -2 >> $O; }
run tools/fuzz_parity.py 150 401 runs
run tools/fuzz_parity.py 150 402
run tools/fuzz_parity.py 200 408 tail
run tools/fuzz_dist.py 80 403
run tools/fuzz_fastx.py 300 405 gpu
run tools/fuzz_cli.py 100 404
run tools/fuzz_parity.py 40 406 inexact
run tools/fuzz_chains.py 40 407
run tools/fuzz_reuse.py 60 409
cat $O
