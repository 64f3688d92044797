#!/bin/bash
# the stage wall at the benched configuration with its laps: FASTA of N reads (disco_amd/bin/readgen) through disco_amd/bin/buildG
# usage: tools/stage_probe.sh [N=50000000] [extra buildG arguments...]
N=${1:-50000000}; shift
D=$(mktemp -d /tmp/disco_stage_XXXX)
disco_amd/bin/readgen $D/reads.fasta $N 150 30.0 42 150 5000000 > /dev/null 2>&1
echo "MinOverlap4BuildGraph = 40" > $D/disco.cfg
for rep in 1 2; do
  rm -f $D/g_*
  T0=$(date +%s.%N)
  DISCO_VERBOSE=1 disco_amd/bin/buildG -se $D/reads.fasta -f $D/g -p $D/disco.cfg -t 16 "$@" > $D/out.txt 2> $D/err.txt
  T1=$(date +%s.%N)
  echo "process wall $(echo "$T1 - $T0" | bc) s"
  grep -E "\[disco host\]|\[disco\] " $D/err.txt | grep -v "probe attempt" | head -60
  grep -E "finished in" $D/out.txt
  echo ----
done
ls -la $D | head; rm -rf $D
