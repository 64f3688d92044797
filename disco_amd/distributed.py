"""Multi-GPU BuildGraph: one process per GPU, reads replicated, QUERY reads range-partitioned across ranks.

Replaces the reference's two multi-process variants (design e-1):
  * buildG-MPI    : every rank holds the full dataset + index and works on its own read-id range, gossiping
                    `allMarked` / contained ids with MPI_Isend/MPI_Recv (MPI/OverlapGraph.cpp:218-246,473-506,524-528)
  * buildG-MPIRMA : the same with the hash data behind MPI_Get (RMA/HashTable.cpp:615-708)
with three bulk collectives over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests):
  1. all-reduce(MIN) of the containment keys                       (after the probe)
  2. ragged all-gather of the per-shard adjacency (degrees + rows)  (after edge selection)
  3. ragged all-gather of the per-shard transitive flags            (after marking)
Results stay sharded: rank r emits the edges whose smaller endpoint lies in its range.

The collectives are written against an *engine* protocol so that the exchange logic runs unchanged on CPU tensors under
gloo (tests/test_distributed_gloo.py); HipEngine adapts disco_amd.buildgraph.BuildGraph (device pointers <-> torch
tensors through disco_memcpy_d2d).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

NOKEY = 0x7FFFFFFFFFFFFFFF  # DISCO_NOKEY: positive as int64 so that a signed MIN orders keys correctly


def shard_range(n: int, rank: int, world: int):
    """contiguous read-id range of a rank (like the N/numprocs blocks of MPI/OverlapGraph.cpp:524-528)"""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def allgather_ragged(t: torch.Tensor, group=None):
    """all-gather 1-D tensors of different lengths; returns (concatenation in rank order, list of lengths)"""
    world = dist.get_world_size(group)
    cnt = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    mx = max(max(counts), 1)
    pad = torch.zeros(mx, dtype=t.dtype, device=t.device)
    pad[: t.numel()] = t
    bufs = [torch.empty(mx, dtype=t.dtype, device=t.device) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    out = torch.cat([b[:c] for b, c in zip(bufs, counts)]) if sum(counts) else torch.zeros(0, dtype=t.dtype, device=t.device)
    return out, counts


def distributed_step(engine, group=None, timing=None):
    """one BuildGraph pass over the engine's resident reads, sharded over the ranks of `group`.
    Returns dict(e_pre, e_out_local, e_out, n_contained, asymmetric_pairs). `timing`: optional dict that receives wall
    milliseconds per stage of this rank."""
    import time as _time

    t_prev = [_time.perf_counter()]

    def lap(name):
        if timing is not None:
            if hasattr(engine, "device"):
                torch.cuda.synchronize(engine.device)
            t = _time.perf_counter()
            timing[name] = timing.get(name, 0.0) + (t - t_prev[0]) * 1e3
            t_prev[0] = t

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = engine.num_reads
    lo, hi = shard_range(n, rank, world)
    engine.build_index()              # replicated: ~2 % of the work
    engine.set_query_range(lo, hi)
    engine.probe()
    lap("index+probe")
    # (1) containment: smallest key wins across ranks
    keys = engine.get_keys()
    dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=group)
    engine.set_keys(keys)
    lap("allreduce_keys")
    n_contained = engine.mark_contained()
    engine.select_edges()
    lap("contain+select")
    # (2) adjacency of every shard to everybody: the reduction of node v reads the lists of v's neighbours
    deg, rows = engine.export_adjacency()
    lap("export")
    deg_all, _ = allgather_ragged(deg, group)
    rows_all, _ = allgather_ragged(rows, group)
    lap("allgather_adjacency")
    engine.import_adjacency(deg_all, rows_all)
    lap("import")
    asym = torch.tensor([engine.symmetrize(False)], dtype=torch.int64, device=keys.device)
    dist.all_reduce(asym, op=dist.ReduceOp.SUM, group=group)
    if int(asym.item()):
        # pairs found from one side only (order-dependent regime of the reference): every rank completes all lists
        engine.symmetrize(True)
        engine.merge_extras()
    lap("symmetrize")
    engine.transitive_mark()
    lap("mark")
    # (3) an edge survives only if it is flagged from neither end -> everybody needs everybody's flags
    flags_local, slot_lo, slot_hi, total = engine.get_flags()
    flags_all, counts = allgather_ragged(flags_local, group)
    assert flags_all.numel() == total, (flags_all.numel(), total)
    engine.set_flags(flags_all)
    lap("allgather_flags")
    e_out_local = engine.emit_edges()
    lap("emit")
    tot = torch.tensor([e_out_local], dtype=torch.int64, device=keys.device)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    return dict(e_pre=total // 2, e_out_local=e_out_local, e_out=int(tot.item()), n_contained=n_contained,
                asymmetric_pairs=int(asym.item()), range=(lo, hi))


class HipEngine:
    """adapts BuildGraph (C-ABI, device pointers) to the engine protocol with torch CUDA tensors"""

    def __init__(self, g, device):
        self.g = g
        self.device = device

    @property
    def num_reads(self):
        return self.g.num_reads

    def build_index(self):
        self.g.build_index()

    def set_query_range(self, lo, hi):
        self.lo, self.hi = lo, hi
        self.g.set_query_range(lo, hi)

    def probe(self):
        self.g.probe()

    def get_keys(self):
        ptr, n = self.g.contain_keys()
        t = torch.empty(n, dtype=torch.int64, device=self.device)
        self.g.memcpy_d2d(t.data_ptr(), ptr, n * 8)
        return t

    def set_keys(self, t):
        torch.cuda.synchronize(self.device)
        ptr, n = self.g.contain_keys()
        self.g.memcpy_d2d(ptr, t.data_ptr(), n * 8)

    def mark_contained(self):
        return self.g.mark_contained()

    def select_edges(self):
        self.g.select_edges()

    def export_adjacency(self):
        total = self.g.adjacency_size()
        deg = torch.empty(max(self.hi - self.lo, 1), dtype=torch.int32, device=self.device)[: self.hi - self.lo]
        rows = torch.empty(max(total, 1), dtype=torch.int64, device=self.device)[:total]
        self.g.export_adjacency(deg.data_ptr(), rows.data_ptr())
        return deg, rows

    def import_adjacency(self, deg_all, rows_all):
        torch.cuda.synchronize(self.device)
        assert deg_all.numel() == self.num_reads
        self.g.import_adjacency(deg_all.data_ptr(), rows_all.data_ptr() if rows_all.numel() else 0, rows_all.numel())

    def symmetrize(self, full):
        return self.g.symmetrize(full)

    def merge_extras(self):
        self.g.merge_extras()

    def transitive_mark(self):
        self.g.transitive_mark()

    def get_flags(self):
        ptr, slot_lo, slot_hi, total = self.g.tr_flags()
        nloc = slot_hi - slot_lo
        t = torch.empty(max(nloc, 1), dtype=torch.uint8, device=self.device)[:nloc]
        if nloc:
            self.g.memcpy_d2d(t.data_ptr(), ptr + slot_lo, nloc)
        return t, slot_lo, slot_hi, total

    def set_flags(self, flags_all):
        torch.cuda.synchronize(self.device)
        ptr, _, _, total = self.g.tr_flags()
        if total:
            self.g.memcpy_d2d(ptr, flags_all.data_ptr(), total)

    def emit_edges(self):
        return self.g.emit_edges()
