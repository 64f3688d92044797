"""CPU, world_size 2 and 3 over gloo: the N>1 exchange logic of disco_amd/distributed.py (the same code that runs over
RCCL on the GPUs) with a mock engine whose shard data is self-describing, so every rank can check that it received the
global min of the containment keys, every adjacency shard at its place of the rank-major padded layout and every node's
survivor list (or everybody's flags in the fallback cases)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from disco_amd import distributed as D  # noqa: E402

N_READS = 1003  # not divisible by 2 or 3: ragged shards


def deg_of(v):
    return (v * 7) % 5  # 0..4 entries per node


def row_of(v):
    return [v * 1000 + t for t in range(deg_of(v))]


class MockEngine:
    def __init__(self, rank, world, asym_on_rank=None, wide_on_rank=None, compact=False, dropped_on_rank=None):
        self.rank, self.world = rank, world
        self.supports_compact = compact
        self.dropped_on_rank = dropped_on_rank
        self.num_reads = N_READS
        self.asym_on_rank = asym_on_rank
        self.wide_on_rank = wide_on_rank
        self.log = []
        self._bufs = {}

    def buffer(self, name, numel, dtype):
        t = self._bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.full((numel,), -7 if dtype != torch.uint8 else 9, dtype=dtype)  # garbage on purpose
            self._bufs[name] = t
        return t[:numel]

    def build_index(self):
        self.log.append("index")

    def set_query_range(self, lo, hi):
        self.lo, self.hi = lo, hi

    def probe(self):
        self.log.append("probe")

    def get_keys(self):
        # every rank proposes a key for every read; rank (i % world) proposes the smallest one; some stay "not contained"
        k = torch.full((N_READS,), D.NOKEY, dtype=torch.int64)
        for i in range(N_READS):
            if i % 4 == 0:
                continue
            k[i] = (i + 1) * 131072 + (0 if i % self.world == self.rank else 1 + self.rank)
        return k

    def set_keys(self, t):
        self.keys = t.clone()

    def mark_contained(self):
        return int((self.keys != D.NOKEY).sum())

    def select_edges(self):
        self.log.append("select")

    def adjacency_size(self):
        return sum(deg_of(v) for v in range(self.lo, self.hi))

    def total_edges(self):
        return sum(deg_of(v) for v in range(N_READS))

    def export_adjacency(self, deg_view, rows_view):
        deg_view.copy_(torch.tensor([deg_of(v) for v in range(self.lo, self.hi)], dtype=torch.int32))
        rows = [x for v in range(self.lo, self.hi) for x in row_of(v)]
        rows_view[: len(rows)] = torch.tensor(rows, dtype=torch.int64)

    def dropped_hits(self):
        return 5 if self.dropped_on_rank == self.rank else 0

    def set_global_dropped(self, n_all):
        self.global_dropped = n_all

    def export_adjacency32(self, deg_view, rows32_view):
        self.log.append("export32")
        deg_view.copy_(torch.tensor([deg_of(v) for v in range(self.lo, self.hi)], dtype=torch.int32))
        rows = [x % 100003 for v in range(self.lo, self.hi) for x in row_of(v)]
        rows32_view[: len(rows)] = torch.tensor(rows, dtype=torch.int32)

    def adopt_neighbours32(self, deg_pad, rows32_pad, per, mx, world):
        self.adopted32 = (deg_pad.clone(), rows32_pad.clone(), per, mx, world)

    def adopt_adjacency(self, deg_pad, rows_pad, per, mx, world):
        self.adopted = (deg_pad.clone(), rows_pad.clone(), per, mx, world)

    def symmetrize(self, full):
        self.log.append("sym_full" if full else "sym")
        if full:  # the full pass returns the count of all ranks' lists
            return 3 if self.asym_on_rank is not None else 0
        return 3 if self.asym_on_rank == self.rank else 0

    def merge_extras(self):
        self.log.append("merge")

    def transitive_mark(self):
        self.log.append("mark")

    def n_wide(self):
        return 2 if self.wide_on_rank == self.rank else 0

    def export_half(self, half_view, hcnt_view):
        self.log.append("half")
        half_view.copy_(torch.tensor([v * 10 + r for v in range(self.lo, self.hi) for r in range(4)], dtype=torch.int64))
        hcnt_view.copy_(torch.tensor([v % 3 for v in range(self.lo, self.hi)], dtype=torch.int32))

    def import_half(self, half_all, hcnt_all):
        self.half_all, self.hcnt_all = half_all.clone(), hcnt_all.clone()

    def get_flags(self):
        self.log.append("flags")
        nloc = self.adjacency_size()
        if "merge" in self.log:  # compact node-ordered layout after the merge
            start = sum(deg_of(v) for v in range(self.lo))
            span = self.total_edges()
        else:                    # rank-major padded layout of the adopted rows
            mx = self.adopted[3]
            start = self.rank * mx
            span = self.world * mx
        return torch.full((nloc,), self.rank + 1, dtype=torch.uint8), start, start + nloc, span

    def set_flags(self, t):
        self.flags_all = t.clone()

    def emit_edges(self):
        return self.hi - self.lo


def _worker_compact(rank, world, port, wide_on_rank, dropped_on_rank, q):
    """the 4-byte row exchange: used iff nobody dropped a hit; falls back to the 8-byte exchange when a rank reports wide nodes"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = MockEngine(rank, world, None, wide_on_rank, compact=True, dropped_on_rank=dropped_on_rank)
        res = D.distributed_step(eng)
        per = (N_READS + world - 1) // world
        shard_tot = [sum(deg_of(v) for v in range(*D.shard_range(N_READS, r, world))) for r in range(world)]
        assert eng.global_dropped == (5 if dropped_on_rank is not None else 0)  # every rank learns the sum over all ranks
        assert res["e_pre"] == sum(shard_tot) // 2 and res["e_out"] == N_READS
        if dropped_on_rank is not None:
            assert "export32" not in eng.log and res["exchange"] == "rows64"
        else:
            deg_pad, rows32_pad, per2, mx, w2 = eng.adopted32
            assert (per2, w2, mx) == (per, world, max(shard_tot))
            assert deg_pad[:N_READS].tolist() == [deg_of(v) for v in range(N_READS)]
            for r in range(world):
                lo, hi = D.shard_range(N_READS, r, world)
                want = [x % 100003 for v in range(lo, hi) for x in row_of(v)]
                assert rows32_pad[r * mx:r * mx + len(want)].tolist() == want
            if wide_on_rank is None:
                assert res["exchange"] == "rows32" and eng.log.count("mark") == 1 and "flags" not in eng.log
                assert eng.half_all.tolist() == [v * 10 + r for v in range(N_READS) for r in range(4)]
            else:  # wide nodes: the marking is redone on the 8-byte rows and the flags travel
                assert res["exchange"] == "rows64" and eng.log.count("mark") == 2 and "flags" in eng.log
                mx64 = eng.adopted[3]
                for r in range(world):
                    assert eng.flags_all[r * mx64:r * mx64 + shard_tot[r]].tolist() == [r + 1] * shard_tot[r]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, f"FAIL: {type(e).__name__}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, asym_on_rank, wide_on_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = MockEngine(rank, world, asym_on_rank, wide_on_rank)
        res = D.distributed_step(eng)
        per = (N_READS + world - 1) // world
        # (1) containment keys: global elementwise min, sentinel preserved
        for i in range(N_READS):
            want = D.NOKEY if i % 4 == 0 else (i + 1) * 131072
            assert int(eng.keys[i]) == want, (i, int(eng.keys[i]), want)
        assert res["n_contained"] == sum(1 for i in range(N_READS) if i % 4)
        # (2) adjacency: degrees node-indexed, rows of rank r in node order at r*mx
        deg_pad, rows_pad, per2, mx, w2 = eng.adopted
        assert (per2, w2) == (per, world)
        shard_tot = [sum(deg_of(v) for v in range(*D.shard_range(N_READS, r, world))) for r in range(world)]
        assert mx == max(shard_tot)
        assert deg_pad[:N_READS].tolist() == [deg_of(v) for v in range(N_READS)]
        assert deg_pad[N_READS:].abs().sum() == 0  # the padding nodes of the last rank have no edges
        for r in range(world):
            lo, hi = D.shard_range(N_READS, r, world)
            want = [x for v in range(lo, hi) for x in row_of(v)]
            assert rows_pad[r * mx:r * mx + len(want)].tolist() == want
        assert res["e_pre"] == sum(shard_tot) // 2
        assert res["e_out"] == N_READS  # sum of the mock's local counts
        # (3) survivors: half lists of every node, or flags in the layout of the rows
        if asym_on_rank is None and wide_on_rank is None:
            assert "flags" not in eng.log and "sym_full" not in eng.log and "merge" not in eng.log
            assert eng.half_all.tolist() == [v * 10 + r for v in range(N_READS) for r in range(4)]
            assert eng.hcnt_all.tolist() == [v % 3 for v in range(N_READS)]
        elif asym_on_rank is not None:  # one rank saw one-sided pairs -> EVERY rank completes all lists, compact flags
            assert eng.log.count("sym_full") == 1 and eng.log.count("merge") == 1 and "half" not in eng.log
            assert res["asymmetric_pairs"] == 3
            want_flags = []
            for r in range(world):
                want_flags += [r + 1] * shard_tot[r]
            assert eng.flags_all.tolist() == want_flags
        else:  # a node with many survivors somewhere -> flags in the padded layout
            assert "half" not in eng.log
            for r in range(world):
                assert eng.flags_all[r * mx:r * mx + shard_tot[r]].tolist() == [r + 1] * shard_tot[r]
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, f"FAIL: {type(e).__name__}: {e}\n{traceback.format_exc()}"))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,asym,wide", [(2, None, None), (3, None, None), (2, 1, None), (3, None, 2)])
def test_distributed_step_over_gloo(world, asym, wide):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, asym, wide, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(world)], results


@pytest.mark.parametrize("world,wide,dropped", [(2, None, None), (3, None, None), (3, 1, None), (2, None, 1)])
def test_distributed_step_compact_exchange_over_gloo(world, wide, dropped):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_compact, args=(r, world, port, wide, dropped, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(world)], results


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 1003):
        for w in (1, 2, 3, 8):
            rs = [D.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
