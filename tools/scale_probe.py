"""one BuildGraph pass at an arbitrary size / length range on one GPU (robustness probe for BASELINE config-5-like shapes):
   python tools/scale_probe.py READS LEN_MIN LEN_MAX [COVERAGE=30] [SKEW=0] [N_CONTIGS]
   env: TWO_PASS=1 (two-pass verify), ERRORS_PPM=n (substitution errors per 10^6 bases), MAX_SUBS=t (inexact overlaps, f-4)"""
import sys, time
sys.path.insert(0, '.')
from disco_amd import buildgraph, readgen
n, lmin, lmax = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cov = float(sys.argv[4]) if len(sys.argv) > 4 else 30.0
genome = int(n * (lmin + lmax) / 2 / cov)
skew = int(sys.argv[5]) if len(sys.argv) > 5 else 0
nc = int(sys.argv[6]) if len(sys.argv) > 6 else max(1, genome // 5_000_000)
spec = readgen.GenSpec.coverage(42, n, lmin, cov, n_contigs=nc, len_max=lmax, skew=skew)
import os
g = buildgraph.BuildGraph(min_overlap=40, device=0, flags=buildgraph.FLAG_TWO_PASS_VERIFY if os.environ.get("TWO_PASS") else 0,
                          max_substitutions=int(os.environ.get("MAX_SUBS", "0")))
g.generate_reads(spec)
if os.environ.get("ERRORS_PPM"):
    g.substitute_bases(7, int(os.environ["ERRORS_PPM"]))
for r in range(2):
    t0 = time.perf_counter(); g.run_graph(); g.synchronize(); t = time.perf_counter() - t0
    c = g.counters()
    print("pass %d: %.1f ms  e_pre %d e_out %d contained %d  overlaps/s %.3g  hbm %.1f GB  big_rows %d cap_bind %d asym %d" % (
        r, t * 1e3, c["e_pre"], c["e_out"], c["n_contained"], c["e_pre"] / t, c["hbm_bytes"] / 1e9, c["big_rows"], c["cap_bind_sites"], c["asymmetric_pairs"]), flush=True)
print({k: round(v, 1) for k, v in g.phase_ms().items() if v > 0.5})
