import os, sys
sys.path.insert(0, '.')
from disco_amd import buildgraph, readgen
n = 50_000_000
spec = readgen.GenSpec.coverage(42, n, 150, 30.0, n_contigs=50)
for m in (23, 21, 19, 17, 15, 23):
    os.environ["DISCO_MINIMIZER_LEN"] = str(m)
    g = buildgraph.BuildGraph(min_overlap=40, device=0)
    g.generate_reads(spec)
    for r in range(2):
        g.run_graph(); g.synchronize()
    ph = g.phase_ms(); c = g.counters()
    print("m", m, {k: round(v, 1) for k, v in ph.items() if v > 1}, "sum %.1f" % sum(ph.values()), c["e_pre"], c["kmer_hits"], flush=True)
    g.close()
