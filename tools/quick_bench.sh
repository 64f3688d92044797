#!/bin/bash
# one short bench run on the GPU box, the line kept under gpurun_out/<name>.json, the numbers that matter printed
# usage: tools/quick_bench.sh NAME [bench.py arguments...]
N=${1:-quick}; shift
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-stage --no-host-to-host --steps 5 --warmup 2 "$@" > gpurun_out/$N.json 2> gpurun_out/$N.err || { tail -5 gpurun_out/$N.err; exit 1; }
python - "$N" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/{sys.argv[1]}.json"))
c = d["config"]
print(f"{sys.argv[1]}: {d['ms_per_step']:.2f} ms/step  e_pre={c['e_pre']} e_out={c['e_out']} contained={c['n_contained']} kmer_hits={c['kmer_hits']}")
print("  ", {k: round(v, 2) for k, v in c["phase_ms_rank0"].items()})
PY
