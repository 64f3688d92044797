"""-m gpu : two classes of rows (include/disco_hip.h, disco_long_rows) — read sets with a few reads of more than 256 bases keep 64-byte
rows for the others; results are the oracle's, bit for bit, and those of the one-stride table."""
import numpy as np
import pytest

from disco_amd import buildgraph
from tests.util import assert_parity, canon_hip, run_hip_reads

pytestmark = pytest.mark.gpu
_COMP = str.maketrans("ACGT", "TGCA")


def mixed_reads(seed, n, smin, smax, cov, long_share, lmin, lmax, genome_len=None):
    """n reads drawn from one random genome, both strands: lengths uniform in [smin, smax] except a share of long ones in [lmin, lmax]"""
    rng = np.random.default_rng(seed)
    mean = (1 - long_share) * (smin + smax) / 2 + long_share * (lmin + lmax) / 2
    G = genome_len or max(int(n * mean / cov), lmax + 1)
    genome = "".join(rng.choice(list("ACGT"), G))
    reads = []
    for _ in range(n):
        L = int(rng.integers(lmin, lmax + 1)) if rng.random() < long_share else int(rng.integers(smin, smax + 1))
        p = int(rng.integers(0, G - L + 1))
        s = genome[p:p + L]
        reads.append(s.translate(_COMP)[::-1] if rng.random() < 0.5 else s)
    return reads


def _run(reads, minovl):
    with buildgraph.BuildGraph(min_overlap=minovl) as g:
        g.upload_ascii(reads)
        g.run_graph()
        return g.fetch_edges(), g.fetch_contained(), g.counters(), g.long_rows


@pytest.mark.parametrize("seed,n,smin,smax,cov,share,lmin,lmax,minovl", [
    (1, 6000, 150, 150, 30.0, 0.03, 300, 600, 40),    # the shape the layout is for: 150 bp with a tail of long reads
    (2, 6000, 100, 250, 30.0, 0.05, 257, 1000, 40),   # mixed short class (256-base compare), long reads up to the widest class
    (3, 5000, 150, 150, 100.0, 0.02, 400, 800, 40),   # 100x: wide rows on both sides
    (4, 5000, 120, 160, 30.0, 0.06, 257, 300, 30),    # the reference's default min-overlap; long reads barely long
    (5, 4000, 150, 250, 40.0, 0.04, 500, 1024, 50),   # windows of 27
    (6, 4000, 150, 150, 30.0, 0.001, 600, 600, 40),   # a handful of long reads
    (7, 4000, 150, 150, 30.0, 0.01, 1500, 5000, 40),  # long reads beyond the probe's LDS row (> 1024 bases): the lists walk global memory
    (8, 3000, 100, 250, 30.0, 0.003, 20000, 30000, 45),  # a few reads near the format's limit of 32767 bases
])
def test_mixed_sets_equal_the_oracle(seed, n, smin, smax, cov, share, lmin, lmax, minovl):
    reads = mixed_reads(seed, n, smin, smax, cov, share, lmin, lmax)
    n_long = sum(len(r) > 256 for r in reads)
    assert n_long > 0
    c = assert_parity(reads, minovl, f"two-class seed{seed}")
    assert c["e_out"] > 0
    he, hr, hc, lr = _run(reads, minovl)
    assert lr == n_long, "the set was expected to take two classes of rows"


def test_long_reads_that_contain_and_duplicate_each_other():
    reads = mixed_reads(11, 3000, 150, 150, 30.0, 0.05, 300, 700)
    longs = [r for r in reads if len(r) > 256]
    extra = longs[:20] + [r.translate(_COMP)[::-1] for r in longs[20:40]] + [r[5:-7] for r in longs[40:60]] + [r[:200] for r in longs[60:80]] + [r[-230:] for r in longs[80:100]]
    assert_parity(reads + extra, 40, "two-class dups")


def test_one_stride_and_two_classes_agree(monkeypatch):
    reads = mixed_reads(12, 5000, 150, 150, 30.0, 0.03, 300, 600)
    e2, r2, c2, lr2 = _run(reads, 40)
    monkeypatch.setenv("DISCO_NO_TWO_CLASS", "1")
    e1, r1, c1, lr1 = _run(reads, 40)
    assert lr2 > 0 and lr1 == 0
    a, b = canon_hip(e1, r1), canon_hip(e2, r2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for key in ("probes", "kmer_hits", "n_contained", "e_pre", "e_out"):
        assert c1[key] == c2[key], key


def test_the_table_comes_back_as_it_was_handed_over():
    reads = mixed_reads(13, 3000, 150, 150, 30.0, 0.03, 300, 600)
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.upload_ascii(reads)
        p0, l0 = g.download_reads()
        s0 = g.stride_words
        g.run_graph()
        assert g.long_rows > 0 and g.stride_words == s0
        p1, l1 = g.download_reads()
        assert np.array_equal(p0, p1) and np.array_equal(l0, l1)
        # a second pass over the re-laid table, and a new set of another shape afterwards
        e1 = g.fetch_edges()
        g.run_graph()
        assert np.array_equal(e1, g.fetch_edges())
        g.upload_ascii(reads[:1000])
        g.run_graph()


def test_sets_that_keep_one_stride(monkeypatch):
    # too many long reads, and a window length without minimizer runs (round 6: only with DISCO_NO_GENERIC_RUNS=1 — every window of up to
    # 64 m-mers has run lists otherwise): the one-stride paths, same results
    for reads, mo, env in ((mixed_reads(14, 3000, 150, 150, 30.0, 0.4, 300, 600), 40, None), (mixed_reads(15, 3000, 150, 150, 30.0, 0.03, 300, 600), 80, "DISCO_NO_GENERIC_RUNS")):
        if env:
            monkeypatch.setenv(env, "1")
        with buildgraph.BuildGraph(min_overlap=mo) as g:
            g.upload_ascii(reads)
            g.run_graph()
            assert g.long_rows == 0
        assert_parity(reads, mo, "one stride kept")


def test_two_classes_with_three_word_kmers():
    """round 6: min-overlap 80 (k = 79, windows of 57 m-mers) has run lists now, so a set with a few long reads gets two classes of rows there
    too — the long class through the three-word variants of its kernels"""
    reads = mixed_reads(15, 3000, 150, 150, 30.0, 0.03, 300, 600)
    with buildgraph.BuildGraph(min_overlap=80) as g:
        g.upload_ascii(reads)
        g.run_graph()
        assert g.long_rows == sum(len(r) > 256 for r in reads) > 0 and g.probe_run_words() > 0
    assert_parity(reads, 80, "two classes, k = 79")


def test_generated_long_tail_on_the_device():
    """the generator's tail of long reads (csrc/readgen.h, DISCO_GEN_LONG_*): device rows = numpy twin, two classes taken, oracle parity"""
    from disco_amd import readgen
    from tests.util import run_oracle_reads

    spec = readgen.GenSpec.coverage(seed=77, n_reads=8000, read_len=150, cov=30.0, long_len=600, long_share=1300)
    reads = readgen.generate_reads(spec)
    n_long = sum(len(r) == 600 for r in reads)
    assert 80 < n_long < 320
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.generate_reads(spec)
        g.run_graph()
        assert g.long_rows == n_long
        he, hr, hc = g.fetch_edges(), g.fetch_contained(), g.counters()
    oe, orows, oc = run_oracle_reads(reads, 40)
    a, b = canon_hip(he, hr), canon_hip(oe, orows)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for key in ("probes", "kmer_hits", "n_contained", "e_pre", "e_out"):
        assert hc[key] == oc[key], key


def test_the_real_reference_s_fixture_with_a_tail_of_long_reads():
    """tests/golden/cases.json: tail_20k — 20 000 reads, 1 % of them 600 bp, outputs of the REAL reference (make_golden.py)"""
    from tests import golden_util as gu

    reads, fidx, mo = gu.case_inputs("tail_20k")
    with buildgraph.BuildGraph(min_overlap=mo) as g:
        g.upload_ascii(reads)
        g.run_graph()
        assert g.long_rows == sum(len(r) > 256 for r in reads) > 100
        ce, cc = canon_hip(g.fetch_edges(), g.fetch_contained(), fidx)
    gu.check_against_golden("tail_20k", ce, cc)


@pytest.mark.parametrize("kind", ["pure", "tail", "outlier", "all_long"])
def test_reads_handed_over_back_to_back(kind):
    """disco_upload_reads_ragged: the reads as the reference keeps them (every read at its own length) — same table, same graph as the
    upload with one stride; a set with a tail of long reads gets its two classes straight from the chunks"""
    if kind == "pure":
        reads = mixed_reads(21, 5000, 150, 150, 30.0, 0.0, 300, 300)
    elif kind == "tail":
        reads = mixed_reads(22, 5000, 100, 250, 30.0, 0.03, 257, 2000)
    elif kind == "outlier":
        rng = np.random.default_rng(23)
        reads = mixed_reads(23, 5000, 150, 150, 30.0, 0.0, 300, 300) + ["".join(rng.choice(list("ACGT"), 32767))]
    else:
        reads = mixed_reads(24, 1500, 300, 700, 25.0, 0.0, 300, 300)
    res = {}
    for ragged in (False, True):
        with buildgraph.BuildGraph(min_overlap=40) as g:
            g.upload_ascii(reads, ragged=ragged)
            lr0, s0 = g.long_rows, g.stride_words
            packed, lens = g.download_reads()
            g.run_graph()
            res[ragged] = (packed, lens, lr0, s0, canon_hip(g.fetch_edges(), g.fetch_contained()), g.counters(), g.long_rows)
    a, b = res[False], res[True]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and a[6] == b[6]
    assert np.array_equal(a[4][0], b[4][0]) and np.array_equal(a[4][1], b[4][1])
    for key in ("probes", "kmer_hits", "n_contained", "e_pre", "e_out"):
        assert a[5][key] == b[5][key], key
    n_long = sum(len(r) > 256 for r in reads)
    assert b[2] == (n_long if kind in ("tail", "outlier") else 0)  # (decided at the upload: the chunks are unpacked per class)
    if kind != "all_long":
        assert_parity(reads, 40, f"ragged {kind}")


def test_back_to_back_upload_checks_lengths_and_chunks(monkeypatch):
    reads = mixed_reads(25, 3000, 100, 250, 30.0, 0.02, 300, 900)
    monkeypatch.setenv("DISCO_UPLOAD_CHUNK", "256")  # twelve chunks, the ring of three staging buffers goes round
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.upload_ascii(reads, ragged=True)
        packed, lens = g.download_reads()
        g.run_graph()
        e1 = canon_hip(g.fetch_edges(), g.fetch_contained())
    monkeypatch.delenv("DISCO_UPLOAD_CHUNK")
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.upload_ascii(reads)
        p2, l2 = g.download_reads()
        g.run_graph()
        e2 = canon_hip(g.fetch_edges(), g.fetch_contained())
        assert np.array_equal(packed, p2) and np.array_equal(lens, l2) and np.array_equal(e1[0], e2[0]) and np.array_equal(e1[1], e2[1])
        with pytest.raises(buildgraph.DiscoError, match="length outside"):
            g.upload_ascii(reads[:10] + ["ACGT" * 5], ragged=True)  # 20 bases <= min-overlap


def test_one_context_through_read_sets_of_every_layout(tmp_path):
    """the same context takes, one after the other: a pure set, a set with a tail (one stride from the host, then back to back), a
    generated set with a tail (re-laid at the index build), a FASTA with a tail (packed per class by the input stage), the pure set
    again — every pass equals a fresh context's"""
    from disco_amd import readgen

    pure = mixed_reads(41, 4000, 150, 150, 30.0, 0.0, 300, 300)
    tail = mixed_reads(42, 4000, 150, 150, 30.0, 0.03, 300, 900)
    spec = readgen.GenSpec.coverage(seed=43, n_reads=4000, read_len=150, cov=30.0, long_len=500, long_share=2000)
    fa = tmp_path / "t.fa"
    fa.write_text("".join(f">r{i}\n{s}\n" for i, s in enumerate(tail)))

    def fresh(load):
        with buildgraph.BuildGraph(min_overlap=40) as g:
            load(g)
            g.run_graph()
            return canon_hip(g.fetch_edges(), g.fetch_contained()), g.long_rows

    steps = [("pure", lambda g: g.upload_ascii(pure)), ("tail, one stride", lambda g: g.upload_ascii(tail)), ("tail, back to back", lambda g: g.upload_ascii(tail, ragged=True)),
             ("generated tail", lambda g: g.generate_reads(spec)), ("fasta tail", lambda g: g.ingest_fasta([str(fa)], threads=2)), ("pure again", lambda g: g.upload_ascii(pure)),
             ("tail again", lambda g: g.upload_ascii(tail)), ("tail, a narrower query range", lambda g: (g.upload_ascii(tail), g.set_query_range(0, len(tail)))[0])]
    want = {name: fresh(load) for name, load in steps}
    with buildgraph.BuildGraph(min_overlap=40) as g:
        for name, load in steps:
            load(g)
            g.run_graph()
            got = canon_hip(g.fetch_edges(), g.fetch_contained())
            assert np.array_equal(got[0], want[name][0][0]) and np.array_equal(got[1], want[name][0][1]), name
            assert g.long_rows == want[name][1], name
            assert (g.long_rows > 0) == ("tail" in name), name
            g.run_graph()  # a second pass over the same table
            again = canon_hip(g.fetch_edges(), g.fetch_contained())
            assert np.array_equal(got[0], again[0]) and np.array_equal(got[1], again[1]), name + " (second pass)"


def test_ten_million_reads_with_a_tail_equal_the_one_stride_pass(monkeypatch):
    """10 M x 150 bp with 0.1 % reads of 600 bp (the verdict's shape at a fifth of BASELINE config 3's size): the pass over two classes of
    rows and the pass over rows of one stride produce the same edges and contained rows (digests of the canonical forms)"""
    import hashlib

    from disco_amd import readgen

    spec = readgen.GenSpec.coverage(42, 10_000_000, 150, 30.0, n_contigs=10, long_len=600, long_share=66)

    def run():
        with buildgraph.BuildGraph(min_overlap=40) as g:
            g.generate_reads(spec)
            g.run_graph()
            ce, cc = canon_hip(g.fetch_edges(), g.fetch_contained())
            return hashlib.sha256(np.ascontiguousarray(ce).tobytes()).hexdigest(), hashlib.sha256(np.ascontiguousarray(cc).tobytes()).hexdigest(), g.counters(), g.long_rows

    e2, c2, k2, l2 = run()
    monkeypatch.setenv("DISCO_NO_TWO_CLASS", "1")
    e1, c1, k1, l1 = run()
    assert l2 > 5000 and l1 == 0
    assert e1 == e2 and c1 == c2
    for key in ("probes", "kmer_hits", "n_contained", "e_pre", "e_out", "cap_bind_sites", "asymmetric_pairs"):
        assert k1[key] == k2[key], key
    assert k2["cap_bind_sites"] == 0 and k2["asymmetric_pairs"] == 0  # (inside the order-independent domain: the reference's result too)


def test_substitutions_are_refused_on_an_uploaded_table_with_two_classes():
    reads = mixed_reads(51, 2000, 150, 150, 30.0, 0.03, 300, 600)
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.upload_ascii(reads)
        assert g.long_rows > 0
        with pytest.raises(buildgraph.DiscoError, match="two classes of rows"):
            g.substitute_bases(3, 1000)
        g.run_graph()  # the table is untouched
        assert g.counters()["e_out"] > 0
