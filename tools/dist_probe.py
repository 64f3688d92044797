#!/usr/bin/env python3
"""multi-GPU flow on ONE GPU: G ranks = G host threads over the in-process communicator; prints every rank's exchange
volumes and timings. usage: dist_probe.py G N_READS [len_min len_max cov passes]
env ERRORS_PPM=n: substitution errors per 10^6 bases (the reads then drop hits at chance repeats: regime 2); PARTITIONED_INDEX=1: the index stays
hash-partitioned (lookups to the owners, records back)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco_amd import readgen  # noqa: E402
from tests.dist_util import run_ranks  # noqa: E402

G, n = int(sys.argv[1]), int(sys.argv[2])
lmin = int(sys.argv[3]) if len(sys.argv) > 3 else 150
lmax = int(sys.argv[4]) if len(sys.argv) > 4 else lmin
cov = float(sys.argv[5]) if len(sys.argv) > 5 else 30.0
passes = int(sys.argv[6]) if len(sys.argv) > 6 else 2
genome = int(n * (lmin + lmax) / 2 / cov)
# LONG_SHARE=s LONG_LEN=l: s / 65536 of the reads have l bases instead (two classes of rows under a communicator; DISCO_DIST_NO_TWO_CLASS=1: one stride)
spec = readgen.GenSpec.coverage(42, n, lmin, cov, n_contigs=max(1, genome // 5_000_000), len_max=lmax, long_len=int(os.environ.get("LONG_LEN", "600")) if os.environ.get("LONG_SHARE") else 0,
                                long_share=int(os.environ.get("LONG_SHARE", "0")))
t0 = time.time()
ppm = int(os.environ.get("ERRORS_PPM", "0"))


def setup(g):
    g.dist_generate_reads(spec)
    if ppm:
        g.substitute_bases(7, ppm)


edges, rows, info, infos = run_ranks(G, 40, setup, passes=passes, partitioned_index=bool(os.environ.get("PARTITIONED_INDEX")))
print(f"kernel ms over the ranks, last pass: {sum(i['kernel_ms'] for i in infos):.1f}; pass wall of rank 0: {info['ms_total']:.1f} ms")
print(f"wall {time.time() - t0:.2f} s; e_pre {info['e_pre']} e_out {info['e_out']} contained {info['n_contained']} regime {info['regime']} "
      f"tr_rounds {info['tr_rounds']} deferred {info['tr_deferred']} asymmetric_pairs {info['asymmetric_pairs']} dropped {info['dropped_hits']}")
for i in infos:
    print(json.dumps({"rank": i["rank"], "ms_total": round(i["ms_total"], 2), "MB_sent": {k: round(v / 1e6, 2) for k, v in i["bytes_sent"].items()},
                      "ms": {k: round(v, 2) for k, v in i["ms"].items()}}))
