"""-m gpu : the sharded (multi-GPU) flow on ONE GPU — G contexts, each owning a query range, run disco_amd.distributed.
distributed_step concurrently (one thread per simulated rank) over an in-process communicator that implements the same
collectives RCCL provides. The canonical output must be identical for G = 1, 2, 3, 4 (SURVEY.md §8e 'parity across GPU
counts'). The torch.distributed glue itself is covered by tests/test_distributed_gloo.py."""
import threading

import numpy as np
import pytest
import torch

from disco_amd import buildgraph, distributed
from tests import golden_util as gu
from tests.util import canon_hip

pytestmark = pytest.mark.gpu


class ThreadComm:
    """lock-step collectives between the threads of one process (all tensors live on the same GPU)"""

    def __init__(self, shared, rank):
        self.s, self.rank, self.world = shared, rank, shared["world"]

    def _exchange(self, item):
        s = self.s
        s["slots"][self.rank] = item
        s["barrier"].wait()
        items = list(s["slots"])
        s["barrier"].wait()
        return items

    def all_reduce(self, t, op):
        torch.cuda.synchronize()
        items = self._exchange(t.clone())
        st = torch.stack(items)
        res = {"min": st.min(dim=0).values, "max": st.max(dim=0).values, "sum": st.sum(dim=0)}[op]
        t.copy_(res)
        torch.cuda.synchronize()

    def allgather_inplace(self, buf, chunk):
        torch.cuda.synchronize()
        items = self._exchange(buf[self.rank * chunk:(self.rank + 1) * chunk].clone())
        for r, it in enumerate(items):
            buf[r * chunk:(r + 1) * chunk] = it
        torch.cuda.synchronize()

    def allgather_ragged(self, t):
        torch.cuda.synchronize()
        items = self._exchange(t.clone())
        return torch.cat(items), [int(i.numel()) for i in items]


def run_sharded(reads, min_overlap, G):
    dev = torch.device("cuda", 0)
    gs = [buildgraph.BuildGraph(min_overlap=min_overlap, device=0) for _ in range(G)]
    engines = [distributed.HipEngine(g, dev) for g in gs]
    shared = dict(world=G, slots=[None] * G, barrier=threading.Barrier(G))
    results, errors = [None] * G, []

    def work(r):
        try:
            gs[r].upload_ascii(reads)
            results[r] = distributed.distributed_step(engines[r], comm=ThreadComm(shared, r))
        except Exception as e:  # pragma: no cover
            errors.append((r, repr(e)))
            shared["barrier"].abort()

    try:
        th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        edges = np.concatenate([g.fetch_edges() for g in gs])
        rows = gs[0].fetch_contained()
        assert sum(r["e_out_local"] for r in results) == len(edges) == results[0]["e_out"]
        run_sharded.last_exchange = results[0]["exchange"]
        return edges, rows, results[0]["e_pre"], results[0]["asymmetric_pairs"]
    finally:
        for g in gs:
            g.close()


@pytest.mark.parametrize("name", ["u150_5k", "mixed_4k"])
@pytest.mark.parametrize("G", [1, 2, 3, 4])
def test_sharded_equals_reference(name, G):
    """regular regime: neighbour rows travel as 4-byte entries"""
    reads, fidx, mo = gu.case_inputs(name)
    edges, rows, e_pre, asym = run_sharded(reads, mo, G)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden(name, ce, cc)
    assert asym == 0 and run_sharded.last_exchange == "rows32"


@pytest.mark.parametrize("G", [2, 3])
def test_sharded_full_row_exchange_equals_reference(G, monkeypatch):
    """the 8-byte exchange (what reads >= 2^30, dropped hits or wide nodes fall back to), forced"""
    monkeypatch.setenv("DISCO_NO_COMPACT", "1")
    reads, fidx, mo = gu.case_inputs("mixed_4k")
    edges, rows, e_pre, asym = run_sharded(reads, mo, G)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden("mixed_4k", ce, cc)
    assert asym == 0 and run_sharded.last_exchange == "rows64"


@pytest.mark.parametrize("G", [2, 3])
def test_sharded_order_dependent_regime_equals_unsharded(G):
    """repeats: one-sided pairs -> every rank completes all lists; many-survivor nodes -> flag exchange"""
    from oracle import pyoracle

    reads, fidx, mo = gu.case_inputs("repeats_8k")
    edges, rows, e_pre, asym = run_sharded(reads, mo, G)
    ce, cc = canon_hip(edges, rows, fidx)
    oce, occ, ocnt = pyoracle.oracle_canonical(reads, fidx, mo)
    assert asym == ocnt["asymmetric_pairs"]
    assert e_pre == ocnt["e_pre"]
    assert np.array_equal(cc, occ) and np.array_equal(ce, oce)


def test_sharded_many_survivor_nodes_take_the_flag_exchange(monkeypatch):
    """force the fallback of exchange (3): with the survivor lists disabled every node counts as 'wide'"""
    reads, fidx, mo = gu.case_inputs("mixed_4k")
    monkeypatch.setenv("DISCO_NO_HALF", "1")
    edges, rows, e_pre, asym = run_sharded(reads, mo, 2)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden("mixed_4k", ce, cc)
    assert run_sharded.last_exchange == "rows64"  # the 4-byte attempt was abandoned after the marking


def _repeat_reads(seed, n, lmin, lmax):
    r3 = np.random.default_rng(seed + 1)
    rep = "".join(r3.choice(list("ACGT"), int(r3.integers(60, 400))))
    genome = "".join("".join(r3.choice(list("ACGT"), int(r3.integers(30, 300)))) + rep for _ in range(int(r3.integers(3, 40))))
    comp = str.maketrans("ACGT", "TGCA")
    reads = []
    for _ in range(n):
        L = int(r3.integers(lmin, lmax + 1))
        if L >= len(genome):
            continue
        p0 = int(r3.integers(0, len(genome) - L))
        s0 = genome[p0:p0 + L]
        reads.append(s0.translate(comp)[::-1] if r3.random() < 0.5 else s0)
    return reads


@pytest.mark.parametrize("G", [2, 3, 4])
def test_sharded_notices_a_twin_missing_from_another_ranks_list(G):
    """ONE cap-bound site on rank 0: the read that still holds the edge lives on another rank, whose one-sided pass is
    unbalanced although none of ITS lists lacks a twin — it must report that (tools/fuzz_sharded.py, it1158)"""
    from tests.util import run_hip_reads

    reads = _repeat_reads(906630212, 1387, 60, 75)
    e1, r1, c1 = run_hip_reads(reads, 50)
    assert c1["asymmetric_pairs"] == 1 and c1["cap_bind_sites"] == 1
    edges, rows, e_pre, asym = run_sharded(reads, 50, G)
    assert (asym, e_pre) == (1, c1["e_pre"])
    ce1, cc1 = canon_hip(e1, r1)
    ce2, cc2 = canon_hip(edges, rows)
    assert np.array_equal(ce1, ce2) and np.array_equal(cc1, cc2)


def test_copying_import_of_a_node_ordered_adjacency():
    """disco_export_adjacency + disco_import_adjacency (the copying form of the exchange: one compact node-ordered CSR instead
    of the adopted rank-major padded buffer): two ranks' exports concatenated by hand, both contexts finish on the copy"""
    reads, fidx, mo = gu.case_inputs("mixed_4k")
    dev = torch.device("cuda", 0)
    n = len(reads)
    gs = [buildgraph.BuildGraph(min_overlap=mo, device=0) for _ in range(2)]
    try:
        engines = [distributed.HipEngine(g, dev) for g in gs]
        parts = []
        for r, (g, e) in enumerate(zip(gs, engines)):
            g.upload_ascii(reads)
            lo, hi = distributed.shard_range(n, r, 2)
            e.build_index()
            e.set_query_range(lo, hi)
            e.probe()
        keys = torch.minimum(engines[0].get_keys().clone(), engines[1].get_keys().clone())
        for e in engines:
            e.set_keys(keys.clone())
            e.mark_contained()
            e.select_edges()
            parts.append(e.export_adjacency())
        deg_all = torch.cat([p[0] for p in parts])
        rows_all = torch.cat([p[1] for p in parts])
        total = 0
        for e in engines:
            e.import_adjacency(deg_all, rows_all)
            assert e.symmetrize(False) == 0
            e.transitive_mark()
        # survivors of all nodes: the flags of both ranks in the compact layout (slots of rank 0's nodes first)
        f0, lo0, hi0, span = engines[0].get_flags()
        f1, lo1, hi1, _ = engines[1].get_flags()
        assert (lo0, hi0, lo1, hi1) == (0, int(parts[0][1].numel()), int(parts[0][1].numel()), span)
        flags = torch.cat([f0, f1])
        for e in engines:
            e.set_flags(flags.clone())
            total += e.emit_edges()
        edges = np.concatenate([g.fetch_edges() for g in gs])
        assert total == len(edges)
        ce, cc = canon_hip(edges, gs[0].fetch_contained(), fidx)
        gu.check_against_golden("mixed_4k", ce, cc)
    finally:
        for g in gs:
            g.close()
