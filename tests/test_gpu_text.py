"""-m gpu : disco_format_edges through the C-ABI — the edge lines of saveParGraphToFile (BG/OverlapGraph.cpp:808-867) formatted on
the GPU, against the same lines formatted here from disco_fetch_edges."""
import numpy as np
import pytest

from disco_amd import buildgraph, readgen

pytestmark = pytest.mark.gpu


def _lines(edges, fidx, flag=2):
    out = []
    for e in edges:
        l1, off = int(e["len_src"]), int(e["offset"])
        ovl = l1 - off
        out.append(f"{fidx[int(e['src'])]}\t{fidx[int(e['dst'])]}\t{int(e['orient'])},{ovl},0,0,{l1},{off},{l1 - 1},{int(e['len_dst'])},0,{ovl - 1},NA,{flag}\n")
    return out


@pytest.mark.parametrize("n_files,mapped", [(1, False), (5, False), (3, True), (200, True)])
def test_gpu_text_equals_lines_formatted_from_the_records(n_files, mapped):
    spec = readgen.GenSpec.coverage(seed=41, n_reads=30_000, read_len=90, cov=25.0, n_contigs=9, len_max=300)
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.generate_reads(spec)
        g.run_graph()
        edges = g.fetch_edges()
        files = g.fetch_edge_files(n_files)
        # file indices as the input stage assigns them when records are filtered: increasing, with gaps, up to 11 digits
        fidx = (np.arange(spec.n_reads, dtype=np.uint64) * np.uint64(3 if mapped else 1) + np.uint64(9_999_999_990 if mapped else 1))
        text, off = g.format_edges(n_files, files if n_files > 1 else None, fidx if mapped else None)
    assert off[0] == 0 and off[-1] == len(text) and np.all(np.diff(off.astype(np.int64)) >= 0)
    want = _lines(edges, fidx)
    for t in range(n_files):
        got = text[int(off[t]):int(off[t + 1])].decode().splitlines(keepends=True)
        assert got == [want[i] for i in np.nonzero(files == t)[0]], t   # the file's edges in fetch order
    assert sum(1 for _ in text.decode().splitlines()) == len(edges)


def test_gpu_text_limits():
    spec = readgen.GenSpec.coverage(seed=42, n_reads=3000, read_len=100, cov=20.0)
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.generate_reads(spec)
        with pytest.raises(buildgraph.DiscoError):
            g.format_edges(1)                       # no graph yet
        g.run_graph()
        with pytest.raises(buildgraph.DiscoError):
            g.format_edges(300, g.fetch_edge_files(300))   # one placement pass per file: at most 256
        with pytest.raises(buildgraph.DiscoError):
            g.format_edges(4)                       # several files need the file of every edge
    with buildgraph.BuildGraph(min_overlap=40, max_substitutions=2) as g:
        g.generate_reads(spec)
        g.run_graph()
        with pytest.raises(buildgraph.DiscoError):
            g.format_edges(1)                       # the substitutions column is the host writer's


def test_edge_text_streamed_into_files_equals_the_fetched_text(tmp_path):
    """round 4: disco_write_edge_text — the formatted lines from the device into the caller's open files (pieces through the pinned
    ring, one writer per file and round) — against the text disco_fetch_edge_text returns; and the contained rows sent ahead
    (disco_start_contained_rows) against the rows fetched on demand"""
    import os

    spec = readgen.GenSpec.coverage(seed=43, n_reads=200_000, read_len=100, cov=25.0, n_contigs=7, len_max=250)
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.generate_reads(spec)
        g.build_index()
        g.probe()
        n_cont = g.mark_contained()
        g.start_contained_rows(grouped=True)   # the rows travel while the edges are selected and reduced
        g.build_edges()
        g.transitive_reduce()
        grouped = g.fetch_contained_grouped()
        rows = g.fetch_contained()
        for n_files in (1, 7):
            files = g.fetch_edge_files(n_files)
            text, off = g.format_edges(n_files, files if n_files > 1 else None, None)
            paths = [str(tmp_path / f"e{n_files}_{t}.txt") for t in range(n_files)]
            fds = [os.open(p, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644) for p in paths]
            g.write_edge_text(fds, threads=5)
            for fd in fds:
                os.close(fd)
            for t, p in enumerate(paths):
                assert open(p, "rb").read() == text[int(off[t]):int(off[t + 1])], (n_files, t)
            assert len(text) > 1_000_000 and sum(os.path.getsize(p) for p in paths) == len(text)
    with buildgraph.BuildGraph(min_overlap=40) as g:   # the same rows without the head start
        g.generate_reads(spec)
        g.run_graph()
        rows2 = g.fetch_contained()
    assert n_cont == len(rows) == len(rows2) > 1000 and np.array_equal(rows, rows2)
    assert grouped is not None and len(grouped) == len(rows)
    key = lambda r: np.lexsort((r["contained"], r["j"], r["super"]))
    assert np.array_equal(grouped, rows[key(rows)])
