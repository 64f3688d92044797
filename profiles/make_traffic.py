#!/usr/bin/env python3
"""profiles/<name>_pmc.json (summarize_pmc.py) -> profiles/probe_traffic.json: HBM-side bytes per launch of the hot kernels
(FETCH_SIZE + WRITE_SIZE, KB x 1024, separate --pmc passes), stamped with the fingerprint of the kernel sources they were
measured with — bench.py quotes them only while the sources are unchanged.  usage: make_traffic.py profiles/r02_50M_pmc.json READS"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src, reads = sys.argv[1], int(sys.argv[2])
pmc = json.load(open(src))
kern = {}
for k, v in pmc.items():
    base = k.split("<")[0].strip()
    # (the candidate-generation phase is probe_runs_kernel since round 3, with probe_kernel for the reads it hands over; bench.py names
    # the phase "probe_kernel"; likewise index_runs_kernel / index_count_kernel)
    base = {"probe_runs_kernel": "probe_kernel", "index_runs_kernel": "index_count_kernel"}.get(base, base)
    if base in ("probe_kernel", "verify_kernel", "edge_select_kernel", "transitive_mark_kernel", "index_count_kernel"):
        kern[base] = kern.get(base, 0.0) + (v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0
out = {"reads": reads, "gpus": 1, "kernels": kern, "kernels_sha16": bench.kernels_sha16(),
       "source": f"{os.path.relpath(src, ROOT)}: FETCH_SIZE + WRITE_SIZE (KB) x 1024 from separate --pmc passes (profiles/prof_pmc.sh); row-gather "
                 "kernels taken undoubled (calibration: gather_rows_kernel reports 1.014 x its known bytes, the streaming copy 0.500 x, "
                 "profiles/r01m_50M_pmc.json)"}
old = os.path.join(ROOT, "profiles", "probe_traffic.json")
if os.path.exists(old):
    try:
        out["calibration"] = json.load(open(old)).get("calibration")
    except Exception:
        pass
json.dump(out, open(old, "w"), indent=1)
print(json.dumps(out, indent=1))
