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
