"""differential fuzzing of the multi-GPU flow and of the two-pass verify: random generator settings (the recipe of fuzz_parity.py:
lengths 45-5000, k = 30-64, coverage 3-1500x, contigs, abundance skew, substitution errors, exact / reverse-complement duplicates,
repeat genomes that bind the cap), every case through
   (a) the single-GPU pass,  (b) the single-GPU pass with DISCO_FLAG_TWO_PASS_VERIFY,  (c) G = 2..5 ranks on one GPU (in-process
   transport), two-pass flag at random
and (a) is checked against the CPU oracle (tests.util.assert_parity); (b) and (c) must equal (a) bit for bit.
   python tools/fuzz_dist.py [ITERATIONS=50] [SEED=1] [inexact]
with "inexact": every case gets substitution errors and a random threshold (the f-4 extension): (a) against the oracle's statement
of the rule, (c) against (a); (b) does not apply (the two-pass verify is an exact-mode form)."""
import os
import sys
import time
import traceback

os.environ.setdefault('DISCO_ORDER_MIN_READS', '1')
sys.path.insert(0, '.')
import numpy as np  # noqa: E402

from disco_amd import buildgraph, readgen  # noqa: E402
from tests.dist_util import run_ranks_reads  # noqa: E402
from tests.util import assert_parity, canon_hip, run_hip_reads  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
inexact = len(sys.argv) > 3 and sys.argv[3] == "inexact"
tail = len(sys.argv) > 3 and sys.argv[3] == "tail"  # round 6: reads of up to 256 bases with a tail of long ones — two classes of rows under a communicator
fails = 0
t0 = time.time()
for it in range(iters):
    lmin = int(rng.choice([45, 60, 80, 100, 150, 151, 168, 200, 256, 257, 300, 500, 1000, 1025, 2000]))
    lmax = lmin if rng.random() < 0.3 else int(lmin + rng.integers(1, 2 * lmin))
    mo = int(rng.choice([31, 32, 33, 40, 41, 50, 64, 65, 66, 80, 88, 95]))  # (round 4: k up to 94)
    if mo >= lmin:
        mo = max(31, lmin - 8)
    cov = float(rng.choice([3, 8, 20, 30, 60, 120, 300, 700]))
    n = int(rng.integers(100, 7000))
    if cov >= 300:
        n = min(n, 2000)
    if lmin >= 1000:
        n = min(n, 1000)
        cov = min(cov, 60.0)
    nc = int(rng.integers(1, 6))
    skew = int(rng.random() < 0.3)
    seed = int(rng.integers(1, 1 << 30))
    G = int(rng.integers(2, 6))
    tp = buildgraph.FLAG_TWO_PASS_VERIFY if rng.random() < 0.5 else 0
    part = False
    if rng.random() < 0.5:  # half of the cases in the shape that takes the minimizer runs; half of those with the index kept partitioned
        lmin = int(rng.choice([45, 60, 100, 150, 151, 200, 256]))
        lmax = lmin if rng.random() < 0.3 else int(min(256, lmin + rng.integers(1, 2 * lmin)))
        mo = 40 if lmin > 48 else max(31, lmin - 8)
        part = mo == 40 and rng.random() < 0.5
    long_len = long_share = 0
    if tail:
        lmin = int(rng.choice([45, 60, 100, 150, 151, 200, 256]))
        lmax = lmin if rng.random() < 0.3 else int(min(256, lmin + rng.integers(1, 2 * lmin)))
        mo = int(rng.choice([33, 40, 40, 58, 66, 80])) if lmin > 100 else max(31, lmin - 8)
        part = False
        long_len = int(rng.choice([257, 300, 400, 600, 1000, 1024, 1025, 2000]))
        long_share = int(rng.choice([66, 300, 1300, 3900, 9800]))
        n = min(n, 4000)
        if rng.random() < 0.5:
            os.environ["DISCO_LOOP_ASYNC"] = "1"
        else:
            os.environ.pop("DISCO_LOOP_ASYNC", None)
    label = f"it{it} seed={seed} n={n} len={lmin}-{lmax} mo={mo} cov={cov} nc={nc} skew={skew} G={G} two_pass_in_ranks={tp} partitioned_index={part}" + (
        f" long={long_len} share={long_share}/65536 async={os.environ.get('DISCO_LOOP_ASYNC', '0')}" if tail else "")
    if os.environ.get("FUZZ_VERBOSE"):
        print("start", label, flush=True)
    try:
        spec = readgen.GenSpec.coverage(seed, n, lmin, cov, n_contigs=nc, len_max=lmax, skew=skew, long_len=long_len, long_share=long_share)
        reads = list(readgen.generate_reads(spec))
        if not tail and rng.random() < 0.25:  # repeats: the cap binds, one-sided pairs -> the order-dependent regime (adjacency gathered)
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
            if len(reads) < 8:
                continue
        if rng.random() < 0.2:
            comp = str.maketrans("ACGT", "TGCA")
            reads += [reads[i] if rng.random() < 0.5 else reads[i].translate(comp)[::-1] for i in rng.integers(0, len(reads), max(len(reads) // 10, 1))]
        if inexact:
            err = float(rng.choice([0.001, 0.003, 0.006, 0.012]))
            tsub = int(rng.choice([1, 2, 3, 5, 9]))
            label += f" err={err} tsub={tsub}"
            r2 = np.random.default_rng(seed + 5)
            out = []
            for s_ in reads:
                b = np.frombuffer(s_.encode(), dtype=np.uint8).copy()
                hit = r2.random(len(b)) < err
                b[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[r2.integers(0, 4, int(hit.sum()))]
                out.append(b.tobytes().decode())
            reads = out
            c = assert_parity(reads, mo, label, max_substitutions=tsub)  # (a) vs the oracle
            e1, r1, _ = run_hip_reads(reads, mo, max_substitutions=tsub)
            ce1, cc1 = canon_hip(e1, r1)
            e3, r3_, info, _ = run_ranks_reads(reads, mo, G, max_substitutions=tsub)  # (c)
            ce3, cc3 = canon_hip(e3, r3_)
            assert np.array_equal(cc1, cc3), f"{G} ranks: contained rows differ ({len(cc1)} vs {len(cc3)})"
            assert np.array_equal(ce1, ce3), f"{G} ranks: edges differ ({len(ce1)} vs {len(ce3)})"
            assert info["e_pre"] == c["e_pre"] and info["asymmetric_pairs"] == c["asymmetric_pairs"] and info["cap_bind_sites"] == c["cap_bind_sites"], (info, c)
            print("ok  ", label, "e_pre", c["e_pre"], "e_out", c["e_out"], "contained", c["n_contained"], "asym", c["asymmetric_pairs"], "regime", info["regime"], flush=True)
            continue
        c = assert_parity(reads, mo, label)  # (a) vs the oracle
        e1, r1, _ = run_hip_reads(reads, mo)
        ce1, cc1 = canon_hip(e1, r1)
        e2, r2, c2 = run_hip_reads(reads, mo, flags=buildgraph.FLAG_TWO_PASS_VERIFY)  # (b)
        ce2, cc2 = canon_hip(e2, r2)
        assert np.array_equal(ce1, ce2) and np.array_equal(cc1, cc2), "two-pass verify differs"
        e3, r3_, info, _ = run_ranks_reads(reads, mo, G, flags=tp, partitioned_index=part)  # (c)
        ce3, cc3 = canon_hip(e3, r3_)
        assert np.array_equal(cc1, cc3), f"{G} ranks: contained rows differ ({len(cc1)} vs {len(cc3)})"
        assert np.array_equal(ce1, ce3), f"{G} ranks: edges differ ({len(ce1)} vs {len(ce3)})"
        assert info["e_pre"] == c["e_pre"] and info["asymmetric_pairs"] == c["asymmetric_pairs"] and info["cap_bind_sites"] == c["cap_bind_sites"], (info, c)
        print("ok  ", label, "e_pre", c["e_pre"], "e_out", c["e_out"], "contained", c["n_contained"], "regime", info["regime"], "placement", "loci" if info["placement"] else "id ranges", "rounds", info["tr_rounds"],
              "deferred", info["tr_deferred"], flush=True)
    except Exception as e:
        fails += 1
        print("FAIL", label, repr(e)[:300], flush=True)
        traceback.print_exc()
print(f"{iters - fails}/{iters} ok in {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
