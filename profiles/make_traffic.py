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
PHASE_OF = {"index_runs_kernel": "index", "index_count_kernel": "index", "index_fill_kernel": "index", "index_fill_ordered_kernel": "index", "scan_tile_sums_kernel": "index", "scan_sums_kernel": "index",
            "scan_apply_kernel": "index", "probe_runs_kernel": "probe_kernel", "probe_kernel": "probe_kernel", "verify_flat_kernel": "verify",
            "verify_kernel": "verify", "contain_flags_kernel": "contain", "edge_select_kernel": "select", "edge_select_flat_kernel": "select", "transitive_mark_kernel": "trmark",
            "emit_half_kernel": "emit", "emit_kernel": "emit"}
phases = {}
for k, v in pmc.items():
    ph = PHASE_OF.get(k.split("<")[0].strip())
    if ph is None:
        continue
    # (the scans also serve the grouping: rocprofv3's per-kernel sums cannot tell them apart — a few hundred MB either way)
    f, w = v.get("FETCH_SIZE", 0.0) * 1024.0, v.get("WRITE_SIZE", 0.0) * 1024.0
    p = phases.setdefault(ph, {"fetch": 0.0, "write": 0.0, "kernels": [], "valu": 0.0, "salu": 0.0, "lds": 0.0, "lds_bank_conflict": 0.0})
    p["fetch"] += f
    p["write"] += w
    p["kernels"].append(k)
    # round 6: the instruction side of the same profile — wave-level instructions per launch (SQ_INSTS_*: one count per wavefront
    # instruction), and the extra LDS cycles bank conflicts cost — for the issue roofline of bench.py
    p["valu"] += v.get("SQ_INSTS_VALU", 0.0)
    p["salu"] += v.get("SQ_INSTS_SALU", 0.0)
    p["lds"] += v.get("SQ_INSTS_LDS", 0.0)
    p["lds_bank_conflict"] += v.get("SQ_LDS_BANK_CONFLICT", 0.0)
for p in phases.values():
    p["lo"] = p["fetch"] + p["write"]        # FETCH_SIZE as counted (a pure 64-byte row gather calibrates at 1.014 x)
    p["hi"] = 2.0 * p["fetch"] + p["write"]  # ... doubled (a wide coalesced stream calibrates at 0.500 x)
out = {"reads": reads, "gpus": 1, "phases": phases, "kernels_sha16": bench.kernels_sha16(),
       "source": f"{os.path.relpath(src, ROOT)}: FETCH_SIZE, WRITE_SIZE (KB) x 1024 from separate --pmc passes (profiles/prof_pmc.sh) per phase; lo = fetch + "
                 "write, hi = 2 x fetch + write (calibration: gather_rows_kernel reports 1.014 x its known bytes, the streaming copy 0.500 x: the "
                 "hot kernels mix row gathers and streams, the truth lies between)"}
old = os.path.join(ROOT, "profiles", "probe_traffic.json")
if os.path.exists(old):
    try:
        out["calibration"] = json.load(open(old)).get("calibration")
    except Exception:
        pass
json.dump(out, open(old, "w"), indent=1)
print(json.dumps(out, indent=1))
