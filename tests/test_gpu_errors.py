"""-m gpu : error behaviour of the C ABI (include/disco_hip.h): every misuse comes back as an error code with a message, never as
a crash or a silent wrong answer. The reference's counterparts abort the process (MYEXIT, BG/Common.h:64)."""
import numpy as np
import pytest

from disco_amd import buildgraph, readgen
from disco_amd.buildgraph import BuildGraph, DiscoError

pytestmark = pytest.mark.gpu


def _reads(n=300, L=150):
    return list(readgen.generate_reads(readgen.GenSpec.coverage(3, n, L, 20.0)))


def test_k_limits():
    with pytest.raises(DiscoError, match="unsupported"):
        BuildGraph(min_overlap=96)  # k = 95 > 94 (round 4: k up to 94; rounds 1-3: 64)
    with pytest.raises(DiscoError, match="unsupported"):
        BuildGraph(min_overlap=1)
    for mo in (65, 95):  # k = 64 (two-word k-mers), k = 94: the widest supported
        with BuildGraph(min_overlap=mo) as g:
            g.upload_ascii(_reads(300, 150))
            g.run_graph()
            assert g.counters()["e_pre"] > 0


def test_calls_out_of_order_are_state_errors():
    with BuildGraph(min_overlap=40) as g:
        with pytest.raises(DiscoError, match="no reads"):
            g.build_index()
        g.upload_ascii(_reads())
        with pytest.raises(DiscoError, match="index"):
            g.probe()
        g.build_index()
        with pytest.raises(DiscoError):
            g.mark_contained()
        g.probe()
        with pytest.raises(DiscoError):
            g.build_edges()  # contained flags first
        g.mark_contained()
        with pytest.raises(DiscoError):
            g.transitive_reduce()
        with pytest.raises(DiscoError):
            g.fetch_edges()
        g.build_edges()
        g.transitive_reduce()
        assert len(g.fetch_edges()) == g.counters()["e_out"]
        with pytest.raises(DiscoError):  # the query range is fixed once the probe ran
            g.set_query_range(0, 10)


def test_bad_reads_are_rejected_with_a_message():
    with BuildGraph(min_overlap=40) as g:
        with pytest.raises(DiscoError, match="non-ACGT"):
            g.upload_ascii(["ACGTN" * 30])
        with pytest.raises(DiscoError, match="length outside"):
            g.upload_ascii(["ACGT" * 10] * 3)  # 40 bases: not longer than the minimum overlap (BG/Dataset.cpp:305)
        # the context is still usable
        g.upload_ascii(_reads())
        g.run_graph()
        assert g.counters()["n_reads"] == 300


def test_query_range_must_lie_inside_the_reads():
    with BuildGraph(min_overlap=40) as g:
        g.upload_ascii(_reads())
        with pytest.raises(DiscoError, match="outside"):
            g.set_query_range(10, 301)
        with pytest.raises(DiscoError, match="outside"):
            g.set_query_range(20, 10)
        g.set_query_range(0, 0)  # an empty shard is legal
        g.build_index()
        g.probe()
        assert g.mark_contained() == 0


def test_generator_spec_is_validated():
    with BuildGraph(min_overlap=40) as g:
        with pytest.raises(DiscoError, match="bad spec"):
            g.generate_reads(readgen.GenSpec(seed=1, n_reads=10, contig_len=100, n_contigs=1, len_min=150, len_max=150))
        with pytest.raises(DiscoError, match="bad spec"):
            g.generate_reads(readgen.GenSpec(seed=1, n_reads=10, contig_len=1000, n_contigs=0, len_min=150, len_max=150))
