"""-m gpu : a clock next to the parity suite. Round 5 shipped — for a few commits — counters that cost one atomic per long row on ONE
address (12 ns apiece): invisible at 30x coverage, five times the verify kernel at 100x, and green in every parity test. These checks
compare RATES between two coverages in the same process, so they hold on any box."""
import pytest

from disco_amd import buildgraph, readgen

pytestmark = pytest.mark.gpu


def _pass(n, cov):
    spec = readgen.GenSpec.coverage(42, n, 150, cov, n_contigs=max(1, int(n * 150 / cov) // 5_000_000))
    with buildgraph.BuildGraph(min_overlap=40, device=0) as g:
        g.generate_reads(spec)
        g.run_graph()  # allocations
        g.run_graph()
        g.synchronize()
        return g.phase_ms(), g.counters()


def test_long_rows_cost_what_their_candidates_cost():
    """verify, edge selection and marking per candidate / per edge at 100x (every row beyond 64 hits: the big-row paths' counters, lists,
    variants) against 30x: within a factor that a per-row serialisation would blow through"""
    p30, c30 = _pass(2_000_000, 30.0)
    p100, c100 = _pass(1_000_000, 100.0)
    v30 = p30["verify"] / c30["kmer_hits"]
    v100 = p100["verify"] / c100["kmer_hits"]
    assert v100 < 2.5 * v30, (p30, p100, c30["kmer_hits"], c100["kmer_hits"])
    s30 = (p30["select"] + p30["trmark"]) / c30["e_pre"]
    s100 = (p100["select"] + p100["trmark"]) / c100["e_pre"]
    assert s100 < 3.0 * s30, (p30, p100, c30["e_pre"], c100["e_pre"])


def _phases(n, env, monkeypatch, min_overlap=40, long_share=0):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    try:
        spec = readgen.GenSpec.coverage(42, n, 150, 30.0, n_contigs=max(1, int(n * 150 / 30.0) // 5_000_000), long_len=600 if long_share else 0, long_share=long_share)
        with buildgraph.BuildGraph(min_overlap=min_overlap, device=0) as g:
            g.generate_reads(spec)
            g.run_graph()
            g.run_graph()
            g.synchronize()
            ph, c = g.phase_ms(), g.counters()
            return ph, c, sum(ph.values())
    finally:
        for k in env:
            monkeypatch.delenv(k, raising=False)


def test_fall_back_families_stay_within_their_known_factor(monkeypatch):
    """round 6: every fall-back shape is parity-tested but had no clock. Same reads, same process: the phase a fall-back
    replaces against the phase of the path that ships — the factors are the measured ones with room for a small box,
    far below what a per-row serialisation costs (the 12 ns atomic of round 5 was a factor 5)"""
    n = 3_000_000
    base, cb, _ = _phases(n, {}, monkeypatch)
    # round 2's probe (no run lists): 2.4 x the run walk at 50 M reads
    ph, c, _ = _phases(n, {"DISCO_NO_RUNS": "1"}, monkeypatch)
    assert c["e_out"] == cb["e_out"] and ph["probe_kernel"] < 4.0 * base["probe_kernel"], (base, ph)
    # round 3's verify_kernel (wave per read) and edge_select_kernel: 1.3-1.5 x the flat kernels
    ph, c, _ = _phases(n, {"DISCO_NO_FLAT_VERIFY": "1"}, monkeypatch)
    assert c["e_out"] == cb["e_out"] and ph["verify"] < 2.5 * base["verify"], (base, ph)
    ph, c, _ = _phases(n, {"DISCO_NO_FLAT_SELECT": "1"}, monkeypatch)
    assert c["e_out"] == cb["e_out"] and ph["select"] < 2.5 * base["select"], (base, ph)
    # the pair window of verify switched off (rows per lane)
    ph, c, _ = _phases(n, {"DISCO_VERIFY_CACHE": "0"}, monkeypatch)
    assert c["e_out"] == cb["e_out"] and ph["verify"] < 1.6 * base["verify"], (base, ph)
    # the window length as a run-time value (min-overlap 33: windows of 10) against the instantiation for min-overlap 40, per read
    ph33, c33, _ = _phases(n, {}, monkeypatch, min_overlap=33)
    # (windows of 10 m-mers make 2 / 11 runs per window where windows of 17 make 2 / 18: 1.9 x the lookups per read — measured 12.6 against
    # 6.7 ms at 20 M reads; round 2's probe took 17.0 there)
    assert ph33["probe_kernel"] < 2.6 * base["probe_kernel"] and ph33["index"] < 1.8 * base["index"], (base, ph33)
    ph33b, c33b, _ = _phases(n, {"DISCO_NO_GENERIC_RUNS": "1"}, monkeypatch, min_overlap=33)
    assert c33b["e_out"] == c33["e_out"] and ph33["probe_kernel"] < ph33b["probe_kernel"], (ph33, ph33b)


def test_one_stride_table_against_two_classes_of_rows(monkeypatch):
    """0.1 % reads of 600 bases among 150-base reads: two classes of rows keep the pass at the pure set's time; one stride for everybody
    (DISCO_NO_TWO_CLASS=1) is 1.6 x — and must not be much more"""
    n = 3_000_000
    _, c2, t2 = _phases(n, {}, monkeypatch, long_share=66)
    _, c1, t1 = _phases(n, {"DISCO_NO_TWO_CLASS": "1"}, monkeypatch, long_share=66)
    assert c1["e_out"] == c2["e_out"] and c1["n_contained"] == c2["n_contained"]
    assert t1 < 2.6 * t2, (t1, t2)
