#!/usr/bin/env python3
"""rocprofv3 kernel stats of the 8-rank pass (profiles/prof_dist.sh) next to the single-GPU pass: per kernel, the sum over the ranks,
the single-GPU time, and the job's work inflation. usage: dist_kernel_table.py DIST.csv SINGLE.csv [OUT.json]"""
import csv
import json
import sys


def load(p):
    return {r["Name"].split("(")[0].replace("void ", ""): (int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(p))}


D, S = load(sys.argv[1]), load(sys.argv[2])
# not part of a pass: the ranks' read generation and its checks; the in-process transport's copies (RCCL moves those bytes over xGMI)
SETUP = ("generate_reads_kernel", "validate_len_kernel", "probes_sum_kernel", "__amd_rocclr_copyBuffer", "__amd_rocclr_fillBufferAligned")
rows = sorted(((t, c, S.get(k, (0, 0.0))[1], k) for k, (c, t) in D.items()), reverse=True)
tot = sum(t for t, c, st, k in rows if k not in SETUP)
single = sum(st for k, (sc, st) in S.items() if k not in SETUP)
for t, c, st, k in rows[:40]:
    print(f"{t:8.2f} ms  calls={c:4d}  single={st:7.2f}  {'(setup / transport) ' if k in SETUP else ''}{k[:70]}")
print(f"pass kernels, sum over the ranks: {tot:.2f} ms; single-GPU pass: {single:.2f} ms; work inflation {tot / single:.3f}")
if len(sys.argv) > 3:
    json.dump({"sum_over_ranks_ms": round(tot, 3), "single_gpu_ms": round(single, 3), "work_inflation": round(tot / single, 3),
               "kernels": {k: {"ranks_ms": round(t, 3), "calls": c, "single_ms": round(st, 3)} for t, c, st, k in rows if k not in SETUP and (t > 0.05 or st > 0.05)},
               "excluded": {k: round(t, 3) for t, c, st, k in rows if k in SETUP}}, open(sys.argv[3], "w"), indent=1)
