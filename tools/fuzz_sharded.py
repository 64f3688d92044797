"""differential fuzzing of the sharded flow: G simulated ranks (threads on one GPU, real engine and kernels) vs the unsharded pass.
   python tools/fuzz_sharded.py [ITERATIONS=30] [SEED=1]"""
import os, sys, time, traceback
os.environ.setdefault('DISCO_ORDER_MIN_READS', '1')  # the grouped verify order on every data set, however small
sys.path.insert(0, '.')
import numpy as np
from disco_amd import readgen
from tests.util import canon_hip, run_hip_reads
from tests.test_gpu_sharded import run_sharded

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
t0 = time.time()
for it in range(iters):
    lmin = int(rng.choice([60, 100, 150, 200, 300]))
    lmax = lmin if rng.random() < 0.4 else int(lmin + rng.integers(1, lmin))
    mo = int(rng.choice([31, 40, 50, 65]))
    if mo >= lmin:
        mo = 40
    cov = float(rng.choice([8, 30, 60, 150, 400]))
    n = int(rng.integers(300, 4000))
    G = int(rng.integers(2, 5))
    seed = int(rng.integers(1, 1 << 30))
    label = f"it{it} seed={seed} n={n} len={lmin}-{lmax} mo={mo} cov={cov} G={G}"
    try:
        if rng.random() < 0.3:
            r3 = np.random.default_rng(seed + 1)
            rep = "".join(r3.choice(list("ACGT"), int(r3.integers(60, 400))))
            genome = "".join("".join(r3.choice(list("ACGT"), int(r3.integers(30, 300)))) + rep for _ in range(int(r3.integers(3, 40))))
            comp0 = str.maketrans("ACGT", "TGCA")
            reads = []
            for _ in range(n):
                L = int(r3.integers(lmin, lmax + 1))
                if L >= len(genome):
                    continue
                p0 = int(r3.integers(0, len(genome) - L))
                s0 = genome[p0:p0 + L]
                reads.append(s0.translate(comp0)[::-1] if r3.random() < 0.5 else s0)
            label += " repeats"
        else:
            spec = readgen.GenSpec.coverage(seed, n, lmin, cov, n_contigs=int(rng.integers(1, 4)), len_max=lmax, skew=int(rng.random() < 0.3))
            reads = list(readgen.generate_reads(spec))
        e1, r1, c1 = run_hip_reads(reads, mo)
        ce1, cc1 = canon_hip(e1, r1)
        e2, r2, e_pre, asym = run_sharded(reads, mo, G)
        ce2, cc2 = canon_hip(e2, r2)
        assert np.array_equal(cc1, cc2), "contained rows differ"
        assert np.array_equal(ce1, ce2), f"edges differ ({len(ce1)} vs {len(ce2)})"
        assert e_pre == c1["e_pre"] and asym == c1["asymmetric_pairs"], (e_pre, c1["e_pre"], asym, c1["asymmetric_pairs"])
        print("ok  ", label, run_sharded.last_exchange, "e_pre", e_pre, "asym", asym, flush=True)
    except Exception as e:
        fails += 1
        print("FAIL", label, repr(e)[:300], flush=True)
        traceback.print_exc()
print(f"{iters - fails}/{iters} ok in {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
