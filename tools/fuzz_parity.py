"""differential fuzzing: HIP path (C-ABI) vs the CPU oracle on random generator settings.
   python tools/fuzz_parity.py [ITERATIONS=50] [SEED=1] [inexact | runs]
   with "inexact": every data set gets sequencing errors and a random substitution threshold (the f-4 extension, checked against
   the oracle's statement of the same rule, substitutions per edge included); with "runs": min-overlap 30 / 35 / 40 / 45 / 50 and reads of up to 256 bases
   throughout — the shapes that take the minimizer runs of the index pass (index_runs_kernel / probe_runs_kernel), low-complexity and
   repeat genomes (ties of the window minimum: reads handed to probe_kernel's list pass) more often"""
import os, sys, time, traceback
os.environ.setdefault('DISCO_ORDER_MIN_READS', '1')  # the grouped verify order on every data set, however small
sys.path.insert(0, '.')
import numpy as np
from disco_amd import readgen
from tests.util import assert_parity

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
inexact = len(sys.argv) > 3 and sys.argv[3] == "inexact"
tail = len(sys.argv) > 3 and sys.argv[3] == "tail"  # "runs" shapes with a tail of long reads (257..8000 bases, 0.1-19 %): the two classes of rows
runs = len(sys.argv) > 3 and sys.argv[3] == "runs" or tail
fails = 0
t0 = time.time()
for it in range(iters):
    lmin = int(rng.choice([45, 60, 80, 100, 150, 151, 168, 200, 256, 257, 300, 500, 1000, 1025, 2000, 5000]))
    lmax = lmin if rng.random() < 0.4 else int(lmin + rng.integers(1, 2 * lmin))
    mo = int(rng.choice([31, 32, 33, 40, 41, 50, 64, 65, 66, 80, 88, 95]))  # (round 4: k up to 94)
    if runs:
        lmin = int(rng.choice([45, 60, 80, 100, 128, 150, 151, 167, 168, 200, 250, 256]))
        lmax = lmin if rng.random() < 0.4 else int(min(256, lmin + rng.integers(1, 2 * lmin)))
        mo = int(rng.choice([40, 40, 30, 35, 45, 50]))  # the window lengths index_runs_kernel is built for
    if mo >= lmin:
        mo = max(31, lmin - 8)
    cov = float(rng.choice([3, 8, 20, 30, 60, 120, 300, 700, 1500]))
    n = int(rng.integers(300, 9000))
    if cov >= 300:
        n = min(n, 2500)
    if lmin >= 1000:
        n = min(n, 1200)
        cov = min(cov, 60.0)
    nc = int(rng.integers(1, 6))
    skew = int(rng.random() < 0.3)
    err = float(rng.choice([0, 0, 0, 0.002, 0.01]))
    tsub = 0
    if inexact:
        err = float(rng.choice([0.001, 0.003, 0.006, 0.012]))
        tsub = int(rng.choice([1, 2, 3, 5, 9, 40]))
    seed = int(rng.integers(1, 1 << 30))
    label = f"it{it} seed={seed} n={n} len={lmin}-{lmax} mo={mo} cov={cov} nc={nc} skew={skew} err={err} tsub={tsub}"
    try:
        long_len = int(rng.choice([257, 300, 400, 600, 1000, 1024, 1025, 2000, 8000])) if tail else 0
        long_share = int(rng.choice([66, 300, 1300, 3900, 9800, 12500])) if tail else 0  # (up to 19 %: the limit of the two classes is one in five)
        spec = readgen.GenSpec.coverage(seed, n, lmin, cov, n_contigs=nc, len_max=lmax, skew=skew, long_len=long_len, long_share=long_share)
        reads = list(readgen.generate_reads(spec))
        if tail:
            label += f" long={long_len} share={long_share}/65536"
        if err:
            r2 = np.random.default_rng(seed)
            out = []
            for s in reads:
                b = np.frombuffer(s.encode(), dtype=np.uint8).copy()
                hit = r2.random(len(b)) < err
                b[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[r2.integers(0, 4, int(hit.sum()))]
                out.append(b.tobytes().decode())
            reads = out
        if runs and rng.random() < 0.15:  # low-complexity stretches: the smallest m-mer hash of a window ties
            r4 = np.random.default_rng(seed + 2)
            units = ["AC", "AAT", "ACGT", "A", "AGGC", "ACACG"]
            reads = [(s[:int(len(s) * 0.3)] + (units[int(r4.integers(0, len(units)))] * 200)[:int(len(s) * 0.4)] + s[int(len(s) * 0.7):]) if r4.random() < 0.3 else s for s in reads]
            label += " lowcomplexity"
        if rng.random() < (0.35 if runs else 0.2):  # a genome with repeat copies: duplicate destinations, the per-k-mer cap, one-sided pairs
            r3 = np.random.default_rng(seed + 1)
            rep = "".join(r3.choice(list("ACGT"), int(r3.integers(60, 400))))
            genome = "".join("".join(r3.choice(list("ACGT"), int(r3.integers(30, 300)))) + rep for _ in range(int(r3.integers(3, 40))))
            comp0 = str.maketrans("ACGT", "TGCA")
            reads = []
            for _ in range(n):
                L = int(r3.integers(lmin, lmax + 1))
                if tail and r3.random() < long_share / 65536:
                    L = long_len
                if L >= len(genome):
                    continue
                p0 = int(r3.integers(0, len(genome) - L))
                s0 = genome[p0:p0 + L]
                reads.append(s0.translate(comp0)[::-1] if r3.random() < 0.5 else s0)
            label += " repeats"
        if rng.random() < 0.2:  # exact and reverse-complement duplicates
            comp = str.maketrans("ACGT", "TGCA")
            reads += [reads[i] if rng.random() < 0.5 else reads[i].translate(comp)[::-1] for i in rng.integers(0, len(reads), len(reads) // 10)]
        c = assert_parity(reads, mo, label, max_substitutions=tsub)
        print("ok  ", label, "e_pre", c["e_pre"], "e_out", c["e_out"], "contained", c["n_contained"], "cap", c["cap_bind_sites"], "asym", c["asymmetric_pairs"], flush=True)
    except Exception as e:
        fails += 1
        print("FAIL", label, repr(e)[:300], flush=True)
        traceback.print_exc()
print(f"{iters - fails}/{iters} ok in {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
