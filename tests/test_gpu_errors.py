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


def _pack(reads, stride):
    from disco_amd import buildgraph as bg

    L = bg.load()
    packed = np.zeros((len(reads), stride), dtype=np.uint64)
    for i, r in enumerate(reads):
        assert L.disco_pack_ascii(r.encode(), len(r), packed[i].ctypes.data) == 0
    return packed, np.fromiter((len(r) for r in reads), dtype=np.uint16, count=len(reads))


@pytest.mark.parametrize("producer", ["upload", "upload_ragged", "generate"])
def test_adopted_reads_are_never_freed_kept_or_overwritten(producer):
    """disco_adopt_reads hands the context CALLER-owned device buffers. A later upload / generate on the same context — same number of
    reads, same stride: the shape whose table the context would otherwise keep and refill — must neither write into the caller's
    buffers nor take them over (round 4's advisor finding: reads_owned was set before the old table was released)."""
    import torch

    from tests.util import canon_hip

    a = _reads(400, 150)
    spec_b = readgen.GenSpec.coverage(11, 400, 150, 20.0)
    b = list(readgen.generate_reads(spec_b))
    pa, la = _pack(a, 8)
    t_rows, t_len = torch.from_numpy(pa.view(np.int64)).cuda(), torch.from_numpy(la.view(np.int16)).cuda()
    with BuildGraph(min_overlap=40) as g:
        g.adopt_reads(t_rows.data_ptr(), 8, t_len.data_ptr(), 400)
        g.run_graph()
        ra = canon_hip(g.fetch_edges(), g.fetch_contained())
        if producer == "generate":
            g.generate_reads(spec_b)
        else:
            g.upload_ascii(b, ragged=producer == "upload_ragged")
        g.run_graph()
        rb = canon_hip(g.fetch_edges(), g.fetch_contained())
        # the caller's buffers: untouched while the context works on its own table
        assert np.array_equal(t_rows.cpu().numpy().view(np.uint64), pa) and np.array_equal(t_len.cpu().numpy().view(np.uint16), la)
    # ... and alive after the context is gone (a foreign hipFree would have taken them along)
    torch.cuda.synchronize()
    assert np.array_equal(t_rows.cpu().numpy().view(np.uint64), pa)
    for reads, got in ((a, ra), (b, rb)):
        with BuildGraph(min_overlap=40) as f:
            f.upload_ascii(reads)
            f.run_graph()
            want = canon_hip(f.fetch_edges(), f.fetch_contained())
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_a_refused_upload_leaves_no_stale_reads():
    """an upload refused for a bad length drops the previous table and graph: a caller that ignores the error cannot run a graph over
    the reads of the call before (the stride derived from the bad length may itself be out of range: the advisor's second finding)"""
    with BuildGraph(min_overlap=40) as g:
        g.upload_ascii(_reads())
        g.run_graph()
        words = np.zeros(1250, dtype=np.uint64)
        with pytest.raises(DiscoError, match="length outside"):
            g.upload_reads_ragged(words, np.array([40000], dtype=np.uint16))  # beyond 32767: the ragged stride would be 1250 words
        with pytest.raises(DiscoError, match="no reads"):
            g.build_index()
