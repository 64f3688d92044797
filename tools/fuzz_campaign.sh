cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06}_fuzz.txt; : > $O
run() { echo "== $*" >> $O; timeout 1500 python "$@" 2>&1 | tail -2 >> $O; }
run tools/fuzz_parity.py 150 501 runs
run tools/fuzz_parity.py 150 502
run tools/fuzz_parity.py 200 508 tail
run tools/fuzz_dist.py 200 503
run tools/fuzz_dist.py 60 510 inexact
run tools/fuzz_dist.py 120 611 tail
run tools/fuzz_fastx.py 300 505 gpu
run tools/fuzz_cli.py 100 504
run tools/fuzz_parity.py 40 506 inexact
run tools/fuzz_chains.py 40 507
run tools/fuzz_reuse.py 60 509
cat $O
