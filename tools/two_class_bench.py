#!/usr/bin/env python3
"""What a tail of long reads costs: the pass over N x 150 bp reads (30x, min-overlap 40) next to the same set with a share of 600 bp
reads, the latter with two classes of rows (the default) and with one stride (DISCO_NO_TWO_CLASS=1, a process of its own).
usage: two_class_bench.py [N=50000000] [OUT=gpurun_out/r04_two_class.json] [SHARE_64K=66] [LONG_LEN=600]"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r04_two_class.json"
share = int(sys.argv[3]) if len(sys.argv) > 3 else 66
long_len = int(sys.argv[4]) if len(sys.argv) > 4 else 600


def one(kind):
    from disco_amd import buildgraph, readgen

    kw = dict(long_len=long_len, long_share=share) if kind != "pure" else {}
    genome = int(n * 150 / 30.0)
    spec = readgen.GenSpec.coverage(42, n, 150, 30.0, n_contigs=max(1, genome // 5_000_000), **kw)
    with buildgraph.BuildGraph(min_overlap=40, device=0) as g:
        g.generate_reads(spec)
        for _ in range(2):
            g.run_graph()
        g.synchronize()
        walls = []
        for _ in range(5):
            t0 = time.perf_counter()
            g.run_graph()
            g.synchronize()
            walls.append((time.perf_counter() - t0) * 1e3)
        c = g.counters()
        return {"kind": kind, "pass_ms": round(sorted(walls)[len(walls) // 2], 3), "pass_ms_all": [round(w, 2) for w in walls], "long_rows": g.long_rows,
                "stride_words": g.stride_words, "phase_ms": {k: round(v, 3) for k, v in g.phase_ms().items()},
                "e_pre": c["e_pre"], "e_out": c["e_out"], "n_contained": c["n_contained"], "kmer_hits": c["kmer_hits"], "hbm_bytes": c["hbm_bytes"]}


if len(sys.argv) > 5:  # child: one configuration, one JSON line
    print(json.dumps(one(sys.argv[5])))
    raise SystemExit(0)
res = {}
for kind, env in (("pure", {}), ("mixed_two_classes", {}), ("mixed_one_stride", {"DISCO_NO_TWO_CLASS": "1"})):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), out, str(share), str(long_len), kind], env={**os.environ, **env}, stdout=subprocess.PIPE, text=True)
    if p.returncode:
        raise SystemExit(f"{kind}: exit {p.returncode}")
    res[kind] = json.loads(p.stdout.strip().splitlines()[-1])
a, b, c = res["pure"], res["mixed_two_classes"], res["mixed_one_stride"]
assert b["long_rows"] > 0 and c["long_rows"] == 0
assert (b["e_pre"], b["e_out"], b["n_contained"], b["kmer_hits"]) == (c["e_pre"], c["e_out"], c["n_contained"], c["kmer_hits"]), "the two layouts must agree"
doc = {"what": f"{n} x 150 bp reads, 30x, min-overlap 40; mixed: {share}/65536 of the reads are {long_len} bp instead; median of 5 passes after 2, one process per row",
       "results": res, "mixed_two_classes_over_pure": round(b["pass_ms"] / a["pass_ms"], 4), "mixed_one_stride_over_pure": round(c["pass_ms"] / a["pass_ms"], 4)}
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: (v if k != "results" else {kk: (vv["pass_ms"], vv["long_rows"], vv["stride_words"]) for kk, vv in v.items()}) for k, v in doc.items()}, indent=1))
for k, v in res.items():
    print(k, v["phase_ms"])
