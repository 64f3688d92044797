#!/usr/bin/env python3
"""print the per-kernel table of a summarised counter profile (profiles/summarize_pmc.py output): tools/pmc_table.py FILE [N]"""
import json
import sys

d = json.load(open(sys.argv[1]))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for n, k in sorted(d.items(), key=lambda x: -x[1]["_ms"])[:top]:
    f, w = k.get("FETCH_SIZE", 0) * 1024 / 1e9, k.get("WRITE_SIZE", 0) * 1024 / 1e9
    hit, miss = k.get("TCC_HIT_sum", 0), k.get("TCC_MISS_sum", 0)
    ms, valu = k["_ms"], k.get("SQ_INSTS_VALU", 0)
    busy = valu / (1024 * 2.4e9 / 4 * ms * 1e-3) if ms else 0
    lanes = k.get("SQ_THREAD_CYCLES_VALU", 0) / max(k.get("SQ_ACTIVE_INST_VALU", 1), 1) / 64
    wc = max(k.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{n[:36]:36s} ms={ms:7.2f} fetch={f:6.1f} write={w:6.1f} GB l2hit={hit / max(hit + miss, 1):.2f} valu={valu / 1e9:6.2f}G salu={k.get('SQ_INSTS_SALU', 0) / 1e9:5.2f}G "
          f"lds={k.get('SQ_INSTS_LDS', 0) / 1e9:5.2f}G busy={busy:.2f} lanes={lanes:.2f} waves={int(k.get('SQ_WAVES', 0))} vgpr={k.get('_vgpr')} ldsB={k.get('_lds')} "
          f"wait_any={k.get('SQ_WAIT_ANY', 0) / wc:.2f} wait_inst={k.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} bankconf={k.get('SQ_LDS_BANK_CONFLICT', 0) / 1e9:.2f}G")
