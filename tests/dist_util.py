"""run the multi-GPU BuildGraph flow (disco_dist_run_graph) with G ranks = G host threads of this process, every rank a context
of its own on ONE GPU over the in-process communicator (disco_comm_init_local). RCCL refuses two ranks on one device; the
flow above the transport is the same code either way."""
from __future__ import annotations

import threading

import numpy as np

from disco_amd import buildgraph


def run_ranks(G, min_overlap, setup, device=0, gather_reads=True, passes=1, flags=0, max_substitutions=0, subs_out=None, partitioned_index=False, inspect=None):
    """setup(g) puts this rank's reads into context g (collective calls allowed). Returns (edges of all ranks, contained
    rows of all ranks, info of rank 0, infos); subs_out: a list that receives the substitutions of the edges, in their order"""
    gs = [buildgraph.BuildGraph(min_overlap=min_overlap, device=device, flags=flags, max_substitutions=max_substitutions) for _ in range(G)]
    buildgraph.BuildGraph.comm_init_local(gs)
    out, errors = [None] * G, []

    def work(r):
        try:
            g = gs[r]
            setup(g)
            for _ in range(passes):
                g.dist_run_graph(gather_reads, partitioned_index)
            out[r] = (g.fetch_edges(), g.fetch_contained(), g.dist_info(), g.fetch_edge_substitutions() if subs_out is not None else None)
            if inspect is not None:  # (a dict: rank -> whatever the caller reads off the rank's context)
                inspect[r] = {"long_rows": g.long_rows, "probe_run_words": g.probe_run_words()}
        except Exception as e:  # pragma: no cover
            errors.append((r, repr(e)))

    try:
        th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=600)
        assert not errors, errors
        assert all(o is not None for o in out), "a rank did not finish"
        edges = np.concatenate([o[0] for o in out])
        rows = np.concatenate([o[1] for o in out])
        infos = [o[2] for o in out]
        if subs_out is not None:
            subs_out.append(np.concatenate([o[3] for o in out]))
        assert sum(i["e_out_local"] for i in infos) == len(edges) == infos[0]["e_out"]
        assert sum(i["n_contained_local"] for i in infos) == len(rows) == infos[0]["n_contained"]
        return edges, rows, infos[0], infos
    finally:
        for g in gs:
            g.close()


def run_ranks_reads(reads, min_overlap, G, **kw):
    return run_ranks(G, min_overlap, lambda g: g.dist_upload_ascii(reads), **kw)
