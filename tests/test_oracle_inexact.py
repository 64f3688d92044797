"""CPU: the oracle's inexact-overlap EXTENSION (oracle_build_graph_inexact) against an independent statement of
its rule. The reference has no such mode (it compares exactly and writes 0 substitutions, BG/OverlapGraph.cpp:815-816), so
nothing of the reference can pin it; what pins it is (1) threshold 0 IS the pinned restatement, and (2) a brute-force
enumeration written from the rule's definition — placements of one read against another, seeds looked up in a dictionary — that
shares no code with the C file."""
import numpy as np
import pytest

from disco_amd import readgen
from oracle import pyoracle


def _mutated(seed, n, lmin, lmax, cov, rate):
    spec = readgen.GenSpec.coverage(seed, n, lmin, cov, len_max=lmax)
    codes, off = readgen.generate_codes(spec)
    rng = np.random.default_rng(seed + 1000)
    m = rng.random(len(codes)) < rate
    codes = codes.copy()
    codes[m] = (codes[m] + rng.integers(1, 4, int(m.sum())).astype(np.uint8)) % 4
    return codes, off


def _brute(reads, k, t):
    """the rule, from its definition. A candidate = (A, s2 = B or revcomp(B), placement d of s2[0] on A) such that the k-mer at
    one END of s2 lies inside A at a probed position j (0 <= j < len(A) - k) and equals A[j:j+k] exactly. Seeded by s2's first
    k-mer (j = d) the candidate is a containment if s2 ends inside A and otherwise an overlap across A's right end (needs j >= 1);
    seeded by s2's last k-mer (j = d + len(s2) - k) it is a containment if d >= 0 and an overlap across A's left end if d <= 0
    (j >= 1). It counts if the aligned region differs in at most t bases. A read is contained iff some candidate says so with a
    longer container (or an equally long one of smaller id); edges join reads that are not contained."""
    comp = str.maketrans("ACGT", "TGCA")
    n = len(reads)
    ends = {}  # k-mer -> [(B, rev, which end of s2)]
    for b, s in enumerate(reads):
        rc = s.translate(comp)[::-1]
        for rev, s2 in ((0, s), (1, rc)):
            ends.setdefault(s2[:k], []).append((b, rev, 0))
            ends.setdefault(s2[-k:], []).append((b, rev, 1))
    oriented = [(s, s.translate(comp)[::-1]) for s in reads]

    def subs(a, s2, d):
        x0, x1 = max(d, 0), min(len(a), d + len(s2))
        return sum(1 for x in range(x0, x1) if a[x] != s2[x - d])

    contained = set()
    overlaps = {}  # (A, B, rev, d) -> substitutions, A's point of view
    for a_id, a in enumerate(reads):
        for j in range(0, len(a) - k):
            for b, rev, end in ends.get(a[j:j + k], ()):
                if b == a_id:
                    continue
                s2 = oriented[b][rev]
                d = j if end == 0 else j + k - len(s2)
                if end == 0:
                    is_cont = len(a) - j >= len(s2)
                    is_ovl = (not is_cont) and j >= 1
                else:
                    is_cont = d >= 0
                    is_ovl = d <= 0 and j >= 1
                if not (is_cont or is_ovl):
                    continue
                c = subs(a, s2, d)
                if c > t:
                    continue
                if is_cont and (len(a) > len(s2) or (len(a) == len(s2) and a_id < b)):
                    contained.add(b)
                if is_ovl:
                    overlaps[(a_id, b, rev, d)] = c
    return contained, overlaps


@pytest.mark.parametrize("seed,n,lmin,lmax,cov,k,rate,t", [
    (201, 260, 60, 90, 12.0, 30, 0.01, 1),
    (202, 260, 60, 90, 12.0, 30, 0.012, 3),
    (203, 200, 90, 150, 14.0, 39, 0.008, 2),
])
def test_extension_against_the_definition(seed, n, lmin, lmax, cov, k, rate, t):
    codes, off = _mutated(seed, n, lmin, lmax, cov, rate)
    reads = readgen.codes_to_reads(codes, off)
    rows, edges, cnt, esubs = pyoracle.build_graph_inexact(codes, off, k + 1, t)
    assert cnt["cap_bind_sites"] == 0  # the per-k-mer cap is not part of the brute-force statement: keep the data below it
    contained, overlaps = _brute(reads, k, t)
    assert set(int(x) for x in rows["contained"]) == contained
    # pre-reduction graph: undirected pairs of non-contained reads joined by an overlap found from either side
    live = {key: c for key, c in overlaps.items() if key[0] not in contained and key[1] not in contained}
    und = set()
    for (a, b, rev, d) in live:
        # the same physical overlap seen from B: placement of A (or its reverse complement) on B
        la, lb = len(reads[a]), len(reads[b])
        twin = (b, a, rev, -d if rev == 0 else d + lb - la)
        und.add(min((a, b, rev, d), twin))
    assert cnt["e_pre"] == len(und)
    assert cnt["asymmetric_pairs"] == sum(1 for (a, b, rev, d) in live
                                          if (b, a, rev, -d if rev == 0 else d + len(reads[b]) - len(reads[a])) not in live)
    assert cnt["asymmetric_pairs"] > 0 and esubs.max() > 0
    # every emitted edge is one of those overlaps, with the same substitution count
    for e, c in zip(edges, esubs):
        a, b, o, offset = int(e["src"]), int(e["dst"]), int(e["orient"]), int(e["offset"])
        la, lb = int(e["len_src"]), int(e["len_dst"])
        rev = 1 if o in (1, 2) else 0
        d = offset if o >= 2 else (la - offset) - lb  # orient 2,3: s2 starts at offset; 0,1: s2 ends with A's first la - offset bases
        key = (a, b, rev, d)
        twin = (b, a, rev, -d if rev == 0 else d + lb - la)
        assert key in live or twin in live, (key, twin)
        assert (live[key] if key in live else live[twin]) == c


def test_threshold_zero_is_the_restatement():
    codes, off = _mutated(205, 1500, 80, 140, 20.0, 0.004)
    r0, e0, c0 = pyoracle.build_graph(codes, off, 40)
    lib = pyoracle.lib()
    import ctypes as C

    res = pyoracle.Result()
    assert lib.oracle_build_graph_inexact(codes.ctypes.data, off.ctypes.data, len(off) - 1, 40, 0, 0, C.byref(res)) == 0
    assert res.c.e_out == c0["e_out"] and res.c.n_contained == c0["n_contained"] and res.c.e_pre == c0["e_pre"]
    lib.oracle_free_result(C.byref(res))


def test_error_free_reads_are_unchanged_by_the_threshold():
    spec = readgen.GenSpec.coverage(207, 1500, 80, 20.0, len_max=140)
    codes, off = readgen.generate_codes(spec)
    r0, e0, c0 = pyoracle.build_graph(codes, off, 40)
    r2, e2, c2, s2 = pyoracle.build_graph_inexact(codes, off, 40, 3)
    assert np.array_equal(e0, e2) and np.array_equal(np.sort(r0["contained"]), np.sort(r2["contained"])) and not s2.any()
