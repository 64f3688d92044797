set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r01 -- python3 $R/bench.py --reads 50000000 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o r01 -- python3 $R/bench.py --reads 50000000 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o r01 -- python3 $R/bench.py --reads 50000000 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_write.log 2>&1
ls -R $O | head -40
tail -2 $O/bench_trace.log | cut -c1-600
