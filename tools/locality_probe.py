"""upper bound of what a locus-ordered read numbering could buy: the same reads with ids in FILE order (random genome positions,
what a FASTA gives) and with ids sorted by genome position (perfect locality: a read's candidates, neighbours and bitmap words
are the reads next to it). usage: locality_probe.py [N_READS=20000000]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco_amd import buildgraph, readgen  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
genome = int(n * 150 / 30)
spec = readgen.GenSpec.coverage(42, n, 150, 30.0, n_contigs=max(1, genome // 5_000_000))
g = buildgraph.BuildGraph(min_overlap=40)
g.generate_reads(spec)


def run(label):
    for _ in range(3):
        t0 = time.perf_counter()
        g.run_graph()
        g.synchronize()
        t = time.perf_counter() - t0
    c = g.counters()
    print(f"{label}: {t*1e3:.1f} ms  e_pre {c['e_pre']} e_out {c['e_out']} contained {c['n_contained']}  " +
          str({k: round(v, 1) for k, v in g.phase_ms().items() if v > 0.3}), flush=True)


run("file order  ")
packed, lens = g.download_reads()
order = np.empty(n, dtype=np.int64)
for a in range(0, n, 10_000_000):
    gpos, _, _ = readgen.read_locations(spec, a, min(a + 10_000_000, n))
    order[a:a + len(gpos)] = gpos.astype(np.int64)
perm = np.argsort(order, kind="stable")
g.upload_reads(packed[perm], lens[perm])
run("genome order")
os.environ["DISCO_NO_ORDER"] = "1"
run("genome order, no grouping pass")
