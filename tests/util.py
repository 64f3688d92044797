"""shared helpers for the parity tests: run the HIP path through the C-ABI and compare canonical forms with the oracle."""
from __future__ import annotations

import numpy as np

from disco_amd import readgen
from oracle import pyoracle


def canon_hip(edges, rows, file_index=None):
    n_ids = None
    fi = (lambda x: x.astype(np.int64) + 1) if file_index is None else (lambda x: np.asarray(file_index, dtype=np.int64)[x.astype(np.int64)])
    ce = pyoracle.canonical_edges(fi(edges["src"]), fi(edges["dst"]), edges["orient"], edges["offset"], edges["len_src"], edges["len_dst"])
    cc = pyoracle.canonical_contained(fi(rows["contained"]), fi(rows["super"]), rows["orient"], rows["len2"], rows["len1"], rows["start"])
    return ce, cc


def run_hip_reads(reads, min_overlap, **kw):
    from disco_amd import buildgraph

    with buildgraph.BuildGraph(min_overlap=min_overlap, **kw) as g:
        g.upload_ascii(reads)
        g.run_graph()
        return g.fetch_edges(), g.fetch_contained(), g.counters()


def run_oracle_reads(reads, min_overlap, count_hits=True):
    codes, off = pyoracle.encode_reads(reads)
    rows, edges, cnt = pyoracle.build_graph(codes, off, min_overlap, count_hits)
    return edges, rows, cnt


def subs_by_edge(e, subs):
    """(src, dst, orient, offset, substitutions) rows, sorted — both implementations emit an edge from its smaller endpoint"""
    t = np.stack([e["src"].astype(np.int64), e["dst"].astype(np.int64), e["orient"].astype(np.int64), e["offset"].astype(np.int64),
                  np.asarray(subs).astype(np.int64)], axis=1) if len(e) else np.zeros((0, 5), np.int64)
    return t[np.lexsort((t[:, 3], t[:, 2], t[:, 1], t[:, 0]))]


def assert_parity_inexact(reads, min_overlap, max_substitutions, label=""):
    """the inexact-overlap extension against the oracle's statement of the same rule, substitutions per edge included"""
    from disco_amd import buildgraph

    with buildgraph.BuildGraph(min_overlap=min_overlap, max_substitutions=max_substitutions) as g:
        g.upload_ascii(reads)
        g.run_graph()
        he, hr, hc, hs = g.fetch_edges(), g.fetch_contained(), g.counters(), g.fetch_edge_substitutions()
    codes, off = pyoracle.encode_reads(reads)
    orows, oe, oc, osubs = pyoracle.build_graph_inexact(codes, off, min_overlap, max_substitutions)
    ce, cc = canon_hip(he, hr)
    oce, occ = canon_hip(oe, orows)
    assert np.array_equal(cc, occ), f"{label}: contained rows differ ({len(cc)} vs {len(occ)})"
    assert np.array_equal(ce, oce), f"{label}: edge list differs ({len(ce)} vs {len(oce)})"
    for key in ("probes", "n_contained", "e_pre", "e_out", "cap_bind_sites", "asymmetric_pairs"):
        assert hc[key] == oc[key], f"{label}: counter {key}: hip {hc[key]} oracle {oc[key]}"
    assert np.array_equal(subs_by_edge(he, hs), subs_by_edge(oe, osubs)), f"{label}: substitutions per edge differ"
    return hc


def assert_parity(reads, min_overlap, label="", max_substitutions=0):
    if max_substitutions:
        return assert_parity_inexact(reads, min_overlap, max_substitutions, label)
    he, hr, hc = run_hip_reads(reads, min_overlap)
    oe, orows, oc = run_oracle_reads(reads, min_overlap)
    ce, cc = canon_hip(he, hr)
    oce, occ = canon_hip(oe, orows)
    assert np.array_equal(cc, occ), f"{label}: contained rows differ ({len(cc)} vs {len(occ)})"
    assert np.array_equal(ce, oce), f"{label}: edge list differs ({len(ce)} vs {len(oce)})"
    for key in ("probes", "kmer_hits", "n_contained", "e_pre", "e_out", "cap_bind_sites", "asymmetric_pairs"):
        assert hc[key] == oc[key], f"{label}: counter {key}: hip {hc[key]} oracle {oc[key]}"
    return hc
