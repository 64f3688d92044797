"""CPU, world_size 2 and 3 over gloo: the N>1 exchange logic of disco_amd/distributed.py (the same code that runs over
RCCL on the GPUs) with a mock engine whose shard data is self-describing, so every rank can check that it received the
global min of the containment keys, the node-ordered concatenation of all adjacency shards and everybody's flags."""
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
    def __init__(self, rank, world, asym_on_rank=None):
        self.rank, self.world = rank, world
        self.num_reads = N_READS
        self.asym_on_rank = asym_on_rank
        self.log = []

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

    def export_adjacency(self):
        deg = torch.tensor([deg_of(v) for v in range(self.lo, self.hi)], dtype=torch.int32)
        rows = torch.tensor([x for v in range(self.lo, self.hi) for x in row_of(v)], dtype=torch.int64)
        return deg, rows

    def import_adjacency(self, deg_all, rows_all):
        self.deg_all, self.rows_all = deg_all.clone(), rows_all.clone()

    def symmetrize(self, full):
        self.log.append("sym_full" if full else "sym")
        return 3 if (self.asym_on_rank == self.rank and not full) else 0

    def merge_extras(self):
        self.log.append("merge")

    def transitive_mark(self):
        self.log.append("mark")

    def get_flags(self):
        start = sum(deg_of(v) for v in range(self.lo))
        n = sum(deg_of(v) for v in range(self.lo, self.hi))
        total = sum(deg_of(v) for v in range(N_READS))
        return torch.full((n,), self.rank + 1, dtype=torch.uint8), start, start + n, total

    def set_flags(self, t):
        self.flags_all = t.clone()

    def emit_edges(self):
        return self.hi - self.lo


def _worker(rank, world, port, asym_on_rank, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = MockEngine(rank, world, asym_on_rank)
        res = D.distributed_step(eng)
        # (1) containment keys: global elementwise min, sentinel preserved
        for i in range(N_READS):
            want = D.NOKEY if i % 4 == 0 else (i + 1) * 131072
            assert int(eng.keys[i]) == want, (i, int(eng.keys[i]), want)
        assert res["n_contained"] == sum(1 for i in range(N_READS) if i % 4)
        # (2) adjacency: node-ordered concatenation of all shards
        assert eng.deg_all.tolist() == [deg_of(v) for v in range(N_READS)]
        assert eng.rows_all.tolist() == [x for v in range(N_READS) for x in row_of(v)]
        # (3) flags: slot ranges of the ranks in order
        want_flags = []
        for r in range(world):
            lo, hi = D.shard_range(N_READS, r, world)
            want_flags += [r + 1] * sum(deg_of(v) for v in range(lo, hi))
        assert eng.flags_all.tolist() == want_flags
        assert res["e_pre"] == len(want_flags) // 2
        assert res["e_out"] == N_READS  # sum of the mock's local counts
        if asym_on_rank is None:
            assert "sym_full" not in eng.log and "merge" not in eng.log
        else:  # one rank saw one-sided pairs -> EVERY rank completes all lists
            assert eng.log.count("sym_full") == 1 and eng.log.count("merge") == 1
            assert res["asymmetric_pairs"] == 3
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, f"FAIL: {type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,asym", [(2, None), (3, None), (2, 1)])
def test_distributed_step_over_gloo(world, asym):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, asym, q)) for r in range(world)]
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
