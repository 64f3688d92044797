"""-m gpu : BASELINE config 5's shape at FULL size on one GPU — 2 x 10^8 metagenome-like reads of 100-250 bp (100 genomes,
abundances over two orders of magnitude, about 3/4 of the reads contained). No reference run exists at this size (it would
take days); the result is pinned through size-independent properties and through the oracle on a sub-problem:
  * no edge and no containment joins reads of different genomes (random genomes share no 40-mer),
  * the graph restricted to the reads of ONE genome is bit-identical to the oracle run on exactly those reads (reads of other
    genomes can neither add nor remove an edge there),
  * every edge has its twin (the full two-sided search, forced), a second pass reproduces every counter.
The 10 M-read instance of the same generator settings is pinned against the REAL reference (tests/test_gpu_big.py, s100_250_10m)."""
import os

import numpy as np
import pytest

from disco_amd import buildgraph, readgen
from oracle import pyoracle
from tests.util import canon_hip

pytestmark = pytest.mark.gpu


def _host_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except Exception:
        pass
    return 0.0


def test_config5_shape_at_full_size(monkeypatch):
    n = int(os.environ.get("DISCO_CONFIG5_READS", 200_000_000))
    if _host_gb() < 96:  # 1.5 x 10^8 contained rows and 5 x 10^7 edges come back to the host
        n = min(n, 50_000_000)
    spec = readgen.GenSpec.coverage(42, n, 100, 30.0, n_contigs=100, len_max=250, skew=1)
    # first pass: the two-pass verify (what buildG uses on read sets of mixed length); second pass, on a fresh context: the
    # single-pass verify with the full twin search forced — every result counter must agree
    with buildgraph.BuildGraph(min_overlap=40, flags=buildgraph.FLAG_TWO_PASS_VERIFY) as g2:
        g2.generate_reads(spec)
        g2.run_graph()
        c0 = g2.counters()
    with buildgraph.BuildGraph(min_overlap=40) as g:
        g.generate_reads(spec)
        g.run_graph()
        c1 = g.counters()
        monkeypatch.setenv("DISCO_FORCE_TWIN_CHECK", "1")  # the full twin search instead of the "nobody dropped a hit" shortcut
        g.run_graph()
        c2 = g.counters()
        monkeypatch.delenv("DISCO_FORCE_TWIN_CHECK")
        e = g.fetch_edges()
        r = g.fetch_contained()
    for k in ("n_reads", "probes", "kmer_hits", "n_contained", "raw_hits", "e_pre", "e_out", "cap_bind_sites", "asymmetric_pairs"):
        assert c1[k] == c2[k], k
    for k in ("n_reads", "probes", "n_contained", "e_pre", "e_out", "cap_bind_sites", "asymmetric_pairs"):
        assert c0[k] == c1[k], ("two-pass verify", k)
    assert c1["n_reads"] == n and c1["asymmetric_pairs"] == 0 and len(e) == c1["e_out"] and len(r) == c1["n_contained"]
    assert 0.6 * n < c1["n_contained"] < 0.9 * n and c1["e_out"] > 0.1 * n  # the heavy-containment regime
    assert np.all(e["src"] < e["dst"])

    def contig_of(ids):
        out = np.empty(len(ids), dtype=np.int64)
        for a in range(0, len(ids), 20_000_000):
            gpos, _, _ = readgen.read_locations(spec, ids=ids[a:a + 20_000_000])
            out[a:a + 20_000_000] = (gpos // np.uint64(spec.contig_len)).astype(np.int64)
        return out

    ce_src, ce_dst = contig_of(e["src"]), contig_of(e["dst"])
    assert np.array_equal(ce_src, ce_dst), "an edge joins reads of two genomes"
    assert np.array_equal(contig_of(r["contained"]), contig_of(r["super"])), "a read is contained in a read of another genome"
    # one whole genome against the oracle: the least abundant one with enough reads to be a real graph
    per_contig = np.zeros(spec.n_contigs, dtype=np.int64)
    for a in range(0, n, 20_000_000):
        per_contig += np.bincount(contig_of(np.arange(a, min(a + 20_000_000, n), dtype=np.uint64)), minlength=spec.n_contigs)
    ok = np.where((per_contig >= 20_000) & (per_contig <= 400_000))[0]
    assert len(ok), per_contig
    c = int(ok[np.argmin(per_contig[ok])])
    ids = np.concatenate([a + np.nonzero(contig_of(np.arange(a, min(a + 20_000_000, n), dtype=np.uint64)) == c)[0] for a in range(0, n, 20_000_000)]).astype(np.uint64)
    assert len(ids) == per_contig[c]
    codes, off = readgen.generate_codes(spec, ids=ids)
    orows, oedges, ocnt = pyoracle.build_graph(codes, off, 40)
    sub_e = e[ce_src == c]
    sub_r = r[contig_of(r["contained"]) == c]
    local = lambda x: np.searchsorted(ids, x.astype(np.uint64)).astype(np.uint64)  # noqa: E731  read id -> rank among the genome's reads
    for arr, keys in ((sub_e, ("src", "dst")), (sub_r, ("contained", "super"))):
        arr = arr.copy()
        for k in keys:
            arr[k] = local(arr[k])
        if keys[0] == "src":
            sub_e = arr
        else:
            sub_r = arr
    ce, cc = canon_hip(sub_e, sub_r)
    oce, occ = canon_hip(oedges, orows)
    assert len(ce) == ocnt["e_out"] and len(cc) == ocnt["n_contained"]
    assert np.array_equal(cc, occ), "contained rows of the genome differ from the oracle's"
    assert np.array_equal(ce, oce), "edges of the genome differ from the oracle's"
