"""-m gpu : the sharded (multi-GPU) flow of the C-ABI on ONE GPU — G contexts, each owning a query range, with the three
exchanges done in-process. The canonical output must be identical for G = 1, 2, 3, 4 (SURVEY.md §8e 'parity across GPU
counts'). The torch.distributed glue around the same engine calls is covered by tests/test_distributed_gloo.py."""
import numpy as np
import pytest
import torch

from disco_amd import buildgraph, distributed, readgen
from tests import golden_util as gu
from tests.util import canon_hip

pytestmark = pytest.mark.gpu


def run_sharded(reads, min_overlap, G):
    dev = torch.device("cuda", 0)
    gs = [buildgraph.BuildGraph(min_overlap=min_overlap, device=0) for _ in range(G)]
    engines = [distributed.HipEngine(g, dev) for g in gs]
    n = len(reads)
    try:
        for g in gs:
            g.upload_ascii(reads)
        for r, e in enumerate(engines):
            lo, hi = distributed.shard_range(n, r, G)
            e.build_index()
            e.set_query_range(lo, hi)
            e.probe()
        keys = torch.stack([e.get_keys() for e in engines]).min(dim=0).values  # all-reduce(MIN)
        for e in engines:
            e.set_keys(keys)
            e.mark_contained()
            e.select_edges()
        parts = [e.export_adjacency() for e in engines]
        deg_all = torch.cat([p[0] for p in parts])
        rows_all = torch.cat([p[1] for p in parts])
        asym = 0
        for e in engines:
            e.import_adjacency(deg_all, rows_all)
            asym += e.symmetrize(False)
        if asym:
            for e in engines:
                e.symmetrize(True)
                e.merge_extras()
        fl = []
        for e in engines:
            e.transitive_mark()
            fl.append(e.get_flags())
        if asym:  # every rank merged every list: flags of rank r live at its own slot range of the merged CSR
            pass
        flags_all = torch.cat([f[0] for f in fl])
        assert flags_all.numel() == fl[0][3]
        edges, rows = [], None
        for e in engines:
            e.set_flags(flags_all)
            e.emit_edges()
            edges.append(e.g.fetch_edges())
            rows = e.g.fetch_contained()
        cnt = engines[0].g.counters()
        return np.concatenate(edges), rows, fl[0][3] // 2, asym, cnt
    finally:
        for g in gs:
            g.close()


@pytest.mark.parametrize("name", ["u150_5k", "mixed_4k"])
@pytest.mark.parametrize("G", [2, 3, 4])
def test_sharded_equals_reference(name, G):
    reads, fidx, mo = gu.case_inputs(name)
    edges, rows, e_pre, asym, _ = run_sharded(reads, mo, G)
    ce, cc = canon_hip(edges, rows, fidx)
    gu.check_against_golden(name, ce, cc)
    assert asym == 0


def test_sharded_order_dependent_regime_equals_unsharded():
    from oracle import pyoracle

    reads, fidx, mo = gu.case_inputs("repeats_8k")
    edges, rows, e_pre, asym, _ = run_sharded(reads, mo, 3)
    ce, cc = canon_hip(edges, rows, fidx)
    oce, occ, ocnt = pyoracle.oracle_canonical(reads, fidx, mo)
    assert asym == ocnt["asymmetric_pairs"]
    assert e_pre == ocnt["e_pre"]
    assert np.array_equal(cc, occ) and np.array_equal(ce, oce)
