"""-m gpu : disco_contract_chains through the C-ABI — the chains of the reduced graph as composite edges (SURVEY.md §8 f-1,
disco_amd/csrc/disco_chains.h). The files that come out of it are compared with the real parsimplify and with the host walk in
tests/test_host.py; here the records themselves are checked against the edge list they were made from."""
import numpy as np
import pytest

from disco_amd import buildgraph, readgen

pytestmark = pytest.mark.gpu


def _twin(o):
    return ((o >> 1) ^ 1) | (((o & 1) ^ 1) << 1)


def _graph(spec, min_ovl=0, errors_ppm=0):
    g = buildgraph.BuildGraph(min_overlap=40)
    g.generate_reads(spec)
    if errors_ppm:
        g.substitute_bases(5, errors_ppm)
    g.run_graph()
    edges = g.fetch_edges()
    comp, links, absorbed = g.contract_chains(min_ovl)
    return g, edges, comp, links, absorbed


@pytest.mark.parametrize("shape", ["contigs", "metagenome", "errors"])
def test_composite_edges_are_made_of_the_absorbed_edges(shape):
    if shape == "contigs":
        spec = readgen.GenSpec.coverage(seed=21, n_reads=60_000, read_len=150, cov=30.0, n_contigs=3)
    else:
        spec = readgen.GenSpec.coverage(seed=22, n_reads=60_000, read_len=100, cov=25.0, n_contigs=12, len_max=220, skew=1 if shape == "metagenome" else 0)
    min_ovl = 50
    g, edges, comp, links, absorbed = _graph(spec, min_ovl, 2000 if shape == "errors" else 0)
    try:
        kept = (edges["len_src"].astype(np.int64) - edges["offset"]) >= min_ovl
        assert not absorbed[~kept].any()                       # an edge the consumer would not load is in no chain
        assert int(absorbed.sum()) == int(comp["n_links"].sum()) == len(links)
        assert len(comp) > 0 and comp["n_links"].min() >= 2
        if shape == "contigs":
            assert len(comp) <= 3 * 4 and comp["n_links"].max() > 5000   # a contig is a handful of long chains
        # every link is one absorbed edge, in the direction of the walk, and the chain hangs together
        length = np.zeros(spec.n_reads, dtype=np.int64)
        length[edges["src"].astype(np.int64)] = edges["len_src"]
        length[edges["dst"].astype(np.int64)] = edges["len_dst"]
        fwd = {(int(e["src"]), int(e["dst"]), int(e["orient"]), int(e["offset"])): i for i, e in enumerate(edges)}
        deg = np.bincount(np.concatenate([edges["src"][kept], edges["dst"][kept]]).astype(np.int64), minlength=spec.n_reads)
        used = np.zeros(len(edges), dtype=bool)
        for c in comp[np.argsort(-comp["n_links"].astype(np.int64))][:200]:
            at, total = int(c["a"]), 0
            ls = links[int(c["first_link"]):int(c["first_link"]) + int(c["n_links"])]
            for k, l in enumerate(ls):
                to, off, o = int(l["to"]), int(l["offset"]), int(l["orient"])
                i = fwd.get((at, to, o, off))
                if i is None:  # stored the other way round: the twin orientation and the reverse offset
                    i = fwd.get((to, at, _twin(o), off + int(length[to]) - int(length[at])))  # walk offset = len(at) + stored - len(to)
                assert i is not None and absorbed[i] and not used[i], (at, to, o, off)
                used[i] = True
                if k + 1 < len(ls):
                    assert deg[to] == 2                          # inner nodes have exactly two edges
                at, total = to, total + off
            assert at == int(c["b"]) and total == int(c["offset"])
            assert int(c["orient"]) == (int(ls[0]["orient"]) & 2) | (int(ls[-1]["orient"]) & 1)
        # the same records from the edge list handed back by the host (what buildG --gpus N does)
        comp2, links2, absorbed2 = g.contract_chains(min_ovl, edges=edges)
        assert np.array_equal(comp, comp2) and np.array_equal(links, links2) and np.array_equal(absorbed, absorbed2)
    finally:
        g.close()


def test_nothing_to_contract_and_state_errors():
    spec = readgen.GenSpec.coverage(seed=23, n_reads=2000, read_len=100, cov=1.0)   # 1x: hardly any overlaps
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.generate_reads(spec)
        with pytest.raises(buildgraph.DiscoError):
            g.contract_chains(0)                                  # before the graph exists
        g.run_graph()
        comp, links, absorbed = g.contract_chains(0)
        assert int(absorbed.sum()) == len(links) == int(comp["n_links"].sum())
        comp, links, absorbed = g.contract_chains(10_000)         # a filter no overlap passes
        assert len(comp) == 0 and len(links) == 0 and not absorbed.any()
