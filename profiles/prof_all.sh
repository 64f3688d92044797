# the round's evidence in one gpurun call:  gpurun -- 'bash profiles/prof_all.sh r02'
#   gpurun_out/<name>/kernel_stats.csv        rocprofv3 --kernel-trace --stats of `bench.py --reads 50000000 --steps 5`
#   gpurun_out/<name>_50M_pmc.json            the counter passes of profiles/prof_pmc.sh, summarised per kernel
#   gpurun_out/<name>/bench_default.json      the plain default bench run (the line the driver records)
# then, in the repository:  cp … profiles/ ;  python profiles/make_traffic.py profiles/<name>_50M_pmc.json 50000000
N=${1:-r02}
bash profiles/prof_stats.sh $N 50000000 > /dev/null
bash profiles/prof_pmc.sh ${N}_pmc 50000000
python3 profiles/summarize_pmc.py gpurun_out/${N}_pmc gpurun_out/${N}_50M_pmc.json > /dev/null
rm -rf gpurun_out/${N}_pmc
python bench.py > gpurun_out/$N/bench_default.json 2> gpurun_out/$N/bench_default.err
head -8 gpurun_out/$N/kernel_stats.csv
