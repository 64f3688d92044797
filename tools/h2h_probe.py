#!/usr/bin/env python3
"""where the host-to-host graph wall goes (SURVEY.md 8d 't_graph'): upload / pass / fetch of contained rows / fetch of edges, the
fetches into fresh (never touched) and into already-touched result arrays.   python tools/h2h_probe.py [reads] [passes]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco_amd import buildgraph, readgen  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
spec = readgen.GenSpec.coverage(42, n, 150, 30.0, n_contigs=max(1, n // 1_000_000))
with buildgraph.BuildGraph(min_overlap=40) as g:
    g.generate_reads(spec)
    g.run_graph()
    L, h = g.L, g._h
    s = g.stride_words
    dense = int(os.environ.get("H2H_DENSE", "1"))
    p = L.disco_host_alloc(max(n * s * 8, 8))
    lens = np.zeros(n, dtype=np.uint16)
    g._chk(L.disco_download_reads(h, p, lens.ctypes.data))
    src, sw = p, s
    if dense:  # the rows at the words they use (5 of 8 at 150 bp), as a parser would hand them over
        w = int((int(lens.max()) + 31) // 32)
        full = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n, s))
        q = L.disco_host_alloc(max(n * w * 8, 8))
        np.ctypeslib.as_array(C.cast(q, C.POINTER(C.c_uint64)), shape=(n, w))[:] = full[:, :w]
        src, sw = q, w
    rows = edges = None
    for it in range(passes):
        t0 = time.perf_counter()
        g._chk(L.disco_upload_reads(h, src, sw, lens.ctypes.data, n))
        g.synchronize()
        t1 = time.perf_counter()
        g.run_graph()
        g.synchronize()
        t2 = time.perf_counter()
        nc = g._chk(L.disco_fetch_contained(h, None, 0))
        ne = g._chk(L.disco_fetch_edges(h, None, 0))
        fresh = rows is None or it == passes - 1 and os.environ.get("H2H_FRESH_LAST")
        if fresh:
            rows = np.empty(max(nc, 1), dtype=buildgraph.CONTAINED_DTYPE)
            edges = np.empty(max(ne, 1), dtype=buildgraph.EDGE_DTYPE)
        t3 = time.perf_counter()
        g._chk(L.disco_fetch_contained(h, rows.ctypes.data, nc))
        t4 = time.perf_counter()
        g._chk(L.disco_fetch_edges(h, edges.ctypes.data, ne))
        t5 = time.perf_counter()
        print(f"pass {it}: upload {1e3 * (t1 - t0):.1f} ({n * sw * 8 / 1e9:.2f} GB)  graph {1e3 * (t2 - t1):.1f}  contained {1e3 * (t4 - t3):.1f} ({nc} rows)  "
              f"edges {1e3 * (t5 - t4):.1f} ({ne})  total {1e3 * (t2 - t0 + t5 - t3):.1f} ms  [{'fresh' if fresh else 'touched'} result arrays]", flush=True)
