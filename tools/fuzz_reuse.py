"""one context reused for many data sets of different sizes / lengths / parameters (buffers kept across passes must be resized or
rebuilt correctly): every pass is compared with a fresh context.   python tools/fuzz_reuse.py [ITERATIONS=40] [SEED=1]"""
import os, sys
os.environ.setdefault('DISCO_ORDER_MIN_READS', '1')  # the grouped verify order on every data set, however small
sys.path.insert(0, '.')
import numpy as np
from disco_amd import buildgraph, readgen
from tests.util import canon_hip

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
g = buildgraph.BuildGraph(min_overlap=40)
for it in range(iters):
    n = int(rng.choice([50, 400, 3000, 20000, 120000]))
    lmin = int(rng.choice([60, 150, 300]))
    lmax = lmin + int(rng.integers(0, lmin))
    cov = float(rng.choice([10, 30, 200]))
    if cov == 200:
        n = min(n, 20000)
    tailed = lmax <= 256 and rng.random() < 0.5  # a tail of long reads: two classes of rows come and go on the same context
    spec = readgen.GenSpec.coverage(int(rng.integers(1, 1 << 30)), n, lmin, cov, n_contigs=int(rng.integers(1, 4)), len_max=lmax,
                                    long_len=int(rng.choice([300, 700, 2500])) if tailed else 0, long_share=int(rng.choice([100, 1300, 3500])) if tailed else 0)
    mode = int(rng.integers(0, 4))  # 0: generated on the device; 1, 2: uploaded at one stride; 3: uploaded back to back
    try:
        if mode == 0 or n > 20000:
            g.generate_reads(spec)
        else:
            g.upload_ascii(list(readgen.generate_reads(spec)), ragged=mode == 3)
        passes = int(rng.integers(1, 3))
        for _ in range(passes):
            g.run_graph()
        a = canon_hip(g.fetch_edges(), g.fetch_contained())
        ca = g.counters()
        with buildgraph.BuildGraph(min_overlap=40) as f:
            f.generate_reads(spec)
            f.run_graph()
            b = canon_hip(f.fetch_edges(), f.fetch_contained())
            cb = f.counters()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), "results differ"
        for k in ("e_pre", "e_out", "n_contained", "kmer_hits"):
            assert ca[k] == cb[k], (k, ca[k], cb[k])
        print("ok  it%d n=%d len=%d-%d%s cov=%g mode=%d passes=%d e_out=%d long_rows=%d" % (it, n, lmin, lmax, " +tail" if tailed else "", cov, mode, passes, ca["e_out"], g.long_rows), flush=True)
    except Exception as e:
        fails += 1
        print("FAIL it%d n=%d len=%d-%d cov=%g mode=%d: %r" % (it, n, lmin, lmax, cov, mode, e), flush=True)
g.close()
print("%d/%d ok" % (iters - fails, iters))
sys.exit(1 if fails else 0)
