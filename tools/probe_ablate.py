import os, sys
sys.path.insert(0, '.')
from disco_amd import buildgraph, readgen
n = 50_000_000
spec = readgen.GenSpec.coverage(42, n, 150, 30.0, n_contigs=50)
g = buildgraph.BuildGraph(min_overlap=40, device=0)
g.generate_reads(spec)
g.build_index()
for ab in (0, 4, 1, 2, 0):
    os.environ["DISCO_PROBE_ABLATE"] = str(ab)
    for r in range(2):
        g.probe(); g.synchronize()
    ph = g.phase_ms()
    print("ablate", ab, "probe_kernel %.2f verify %.2f" % (ph["probe_kernel"], ph["verify"]), flush=True)
