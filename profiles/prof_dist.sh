# kernel time of the 8-rank pass (ranks serialised on ONE GPU: DISCO_LOOP_SERIALIZE=1) next to the single-GPU pass, one pass each,
# from rocprofv3 --kernel-trace --stats:  gpurun -- 'bash profiles/prof_dist.sh r04'
#   gpurun_out/<name>_dist8_kernel_stats.csv , gpurun_out/<name>_single_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; N=${1:-r04}; READS=${2:-50000000}; O=$R/gpurun_out/${N}_distprof; mkdir -p $O
export DISCO_LOOP_SERIALIZE=1
MODE=dist rocprofv3 --kernel-trace --stats --output-format csv -d $O/dist -o s -- python3 $R/tools/dist_profile.py 8 $READS $O/dist.json > $O/dist.log 2>&1
MODE=single rocprofv3 --kernel-trace --stats --output-format csv -d $O/single -o s -- python3 $R/tools/dist_profile.py 8 $READS $O/single.json > $O/single.log 2>&1
cp $(ls $O/dist/*/*kernel_stats.csv $O/dist/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/${N}_dist8_kernel_stats.csv
cp $(ls $O/single/*/*kernel_stats.csv $O/single/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/${N}_single_kernel_stats.csv
rm -rf $O/dist $O/single
head -30 $R/gpurun_out/${N}_dist8_kernel_stats.csv
