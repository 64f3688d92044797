# per-kernel durations of the bench command (rocprofv3 --kernel-trace --stats), summary copied to gpurun_out/<name>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-stats}; mkdir -p $O
N=${2:-50000000}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o s -- python3 $R/bench.py --reads $N --steps 5 --warmup 1 --no-cpu-baseline --no-stage --no-host-to-host > $O/bench_line.json 2> $O/bench.err
cp $(ls $O/raw/*/*kernel_stats.csv $O/raw/*kernel_stats.csv 2>/dev/null | head -1) $O/kernel_stats.csv
rm -rf $O/raw
head -12 $O/kernel_stats.csv
