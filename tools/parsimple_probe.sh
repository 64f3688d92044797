# stage wall of buildG --par-simple on the benched reads (BASELINE config 3), with the host-side lap timers
# usage: bash tools/parsimple_probe.sh [N_READS=50000000]
N=${1:-50000000}
D=${TMPDIR:-/tmp}/psp_$$
mkdir -p $D/graph $D/assembly
disco_amd/bin/readgen $D/r.fasta $N 150 30 42 > /dev/null
printf 'MinOverlap4BuildGraph = 40\nMinOverlap4SimplifyGraph = 40\n' > $D/disco.cfg
T0=$(date +%s%N)
DISCO_VERBOSE=1 disco_amd/bin/buildG -se $D/r.fasta -f $D/graph/a -p $D/disco.cfg -t 16 --par-simple $D/assembly/a 2>&1 | grep -E "parsimple|Partial|finished in|disco host" 
T1=$(date +%s%N)
echo "stage wall $(( (T1 - T0) / 1000000 )) ms"
ls -la $D/assembly | head -5
rm -rf $D
