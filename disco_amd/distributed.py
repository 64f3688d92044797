"""Multi-GPU BuildGraph: one process per GPU, reads replicated, QUERY reads range-partitioned across ranks.

Replaces the reference's two multi-process variants (design e-1):
  * buildG-MPI    : every rank holds the full dataset + index and works on its own read-id range, gossiping
                    `allMarked` / contained ids with MPI_Isend/MPI_Recv (MPI/OverlapGraph.cpp:218-246,473-506,524-528)
  * buildG-MPIRMA : the same with the hash data behind MPI_Get (RMA/HashTable.cpp:615-708)
with three bulk collectives over RCCL (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests):
  1. all-reduce(MIN) of the containment keys                               (after the probe)
  2. in-place all-gather of the per-shard adjacency (degrees + rows)        (after edge selection); in the regular regime
     (nobody dropped a verified hit, fewer than 2^30 reads) the rows travel as 4-byte (destination, orientation) entries —
     all the marking reads of a neighbour's row — which halves the bytes of the step that bounds the flow on xGMI
  3. in-place all-gather of each node's few surviving edges (32 B / node)   (after marking; one flag byte per adjacency
     slot instead when some node has more than 4 survivors)
Results stay sharded: rank r emits the edges whose smaller endpoint lies in its range.

The collectives are written against an *engine* protocol so that the exchange logic runs unchanged on CPU tensors under
gloo (tests/test_distributed_gloo.py); HipEngine adapts disco_amd.buildgraph.BuildGraph (device pointers <-> torch
tensors through disco_memcpy_d2d).
"""
from __future__ import annotations

import os

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


def allgather_inplace(buf: torch.Tensor, chunk: int, group=None):
    """buf holds world chunks of `chunk` elements, rank r has already written its own data to buf[r*chunk:(r+1)*chunk];
    afterwards every rank holds every chunk. No staging copy: over RCCL this is the in-place all-gather."""
    world = dist.get_world_size(group)
    if world == 1 or chunk == 0:
        return
    rank = dist.get_rank(group)
    mine = buf[rank * chunk:(rank + 1) * chunk]
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(buf[: world * chunk], mine, group=group)
    else:  # gloo (CPU tests)
        dist.all_gather([buf[r * chunk:(r + 1) * chunk] for r in range(world)], mine.clone(), group=group)


class TorchComm:
    """the collectives of distributed_step over torch.distributed (RCCL on the GPUs, gloo in the CPU tests)"""

    _OPS = {"min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX, "sum": dist.ReduceOp.SUM}

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_reduce(self, t, op):
        dist.all_reduce(t, op=self._OPS[op], group=self.group)

    def allgather_inplace(self, buf, chunk):
        allgather_inplace(buf, chunk, self.group)

    def allgather_ragged(self, t):
        return allgather_ragged(t, self.group)


def distributed_step(engine, group=None, timing=None, comm=None):
    """one BuildGraph pass over the engine's resident reads, sharded over the ranks of `group`.
    Returns dict(e_pre, e_out_local, e_out, n_contained, asymmetric_pairs). `timing`: optional dict that receives wall
    milliseconds per stage of this rank.

    Exchange layout: node v belongs to rank v // per (per = ceil(n / world)); per-node arrays (degrees, survivor lists) are
    all-gathered in place as world chunks of `per` nodes, the adjacency rows as world chunks of `mx` entries (mx = the largest
    shard), and the context addresses the gathered rows where they lie (disco_adopt_adjacency) — no staging copies."""
    import time as _time

    t_prev = [_time.perf_counter()]

    def lap(name):
        if timing is not None:
            if hasattr(engine, "device"):
                torch.cuda.synchronize(engine.device)
            t = _time.perf_counter()
            timing[name] = timing.get(name, 0.0) + (t - t_prev[0]) * 1e3
            t_prev[0] = t

    comm = comm or TorchComm(group)
    world, rank = comm.world, comm.rank
    n = engine.num_reads
    per = max((n + world - 1) // world, 1)
    lo, hi = shard_range(n, rank, world)
    nloc = hi - lo
    engine.build_index()              # replicated: ~9 % of a single-GPU pass
    engine.set_query_range(lo, hi)
    engine.probe()
    lap("index+probe")
    # (1) containment: smallest key wins across ranks
    keys = engine.get_keys()
    comm.all_reduce(keys, "min")
    engine.set_keys(keys)
    lap("allreduce_keys")
    n_contained = engine.mark_contained()
    engine.select_edges()
    lap("contain+select")
    # (2) adjacency of every shard to everybody: the reduction of node v reads the lists of v's neighbours
    dev = keys.device
    cnt = torch.tensor([engine.adjacency_size()], dtype=torch.int64, device=dev)
    comm.all_reduce(cnt, "max")
    mx = max(int(cnt.item()), 1)
    tot = torch.tensor([engine.adjacency_size(), engine.dropped_hits()], dtype=torch.int64, device=dev)
    comm.all_reduce(tot, "sum")
    total_entries, dropped_all = int(tot[0].item()), int(tot[1].item())
    # a list can miss a twin only if the twin's owner dropped a verified hit, and the owner may sit on another rank: the
    # shortcut of the twin search needs the count of ALL ranks
    engine.set_global_dropped(dropped_all)
    deg_pad = engine.buffer("deg", world * per, torch.int32)
    deg_pad[rank * per + nloc:(rank + 1) * per].zero_()
    compact = (getattr(engine, "supports_compact", False) and n < (1 << 30) and dropped_all == 0
               and not os.environ.get("DISCO_NO_COMPACT"))
    if compact:
        rows32_pad = engine.buffer("rows32", world * mx, torch.int32)
        engine.export_adjacency32(deg_pad[rank * per:rank * per + nloc], rows32_pad[rank * mx:(rank + 1) * mx])
        lap("export")
        comm.allgather_inplace(deg_pad, per)
        comm.allgather_inplace(rows32_pad, mx)
        lap("allgather_adjacency")
        engine.adopt_neighbours32(deg_pad, rows32_pad, per, mx, world)
        lap("adopt")
        engine.symmetrize(False)  # nothing was dropped anywhere: symmetric by construction, no search
        engine.transitive_mark()
        lap("mark")
        wide = torch.tensor([engine.n_wide()], dtype=torch.int64, device=dev)
        comm.all_reduce(wide, "sum")
        if int(wide.item()) == 0:
            _exchange_half(engine, comm, world, rank, per, nloc, n)
            lap("exchange_survivors")
            e_out_local = engine.emit_edges()
            lap("emit")
            out = torch.tensor([e_out_local], dtype=torch.int64, device=dev)
            comm.all_reduce(out, "sum")
            return dict(e_pre=total_entries // 2, e_out_local=e_out_local, e_out=int(out[0].item()), n_contained=n_contained,
                        asymmetric_pairs=0, range=(lo, hi), exchange="rows32")
        # some node keeps more than 4 edges: its survivors do not fit the 32-byte lists and the flag exchange needs the
        # slots of the full rows -> redo the exchange with 8-byte entries (flags are stripped by the export)
    rows_pad = engine.buffer("rows", world * mx, torch.int64)
    engine.export_adjacency(deg_pad[rank * per:rank * per + nloc], rows_pad[rank * mx:(rank + 1) * mx])
    lap("export")
    comm.allgather_inplace(deg_pad, per)
    comm.allgather_inplace(rows_pad, mx)
    lap("allgather_adjacency")
    engine.adopt_adjacency(deg_pad, rows_pad, per, mx, world)
    lap("adopt")
    asym = torch.tensor([engine.symmetrize(False)], dtype=torch.int64, device=keys.device)
    comm.all_reduce(asym, "sum")
    n_asym = 0
    if int(asym.item()):
        # pairs found from one side only (order-dependent regime of the reference): every rank completes all lists. The
        # per-range counts above only say WHETHER something is missing (a rank whose one-sided pass balances reports 0 even
        # if lists of its range lack twins that the owner of the larger id notices); the full pass returns the count.
        n_asym = engine.symmetrize(True)
        engine.merge_extras()
    lap("symmetrize")
    engine.transitive_mark()
    lap("mark")
    # (3) an edge survives only if it survives the marking of BOTH ends: exchange each node's (few) survivors; only if some
    #     node has more than 4 of them fall back to exchanging one flag byte per adjacency slot
    wide = torch.tensor([engine.n_wide()], dtype=torch.int64, device=keys.device)
    comm.all_reduce(wide, "sum")
    if int(wide.item()) == 0 and int(asym.item()) == 0:
        _exchange_half(engine, comm, world, rank, per, nloc, n)
    else:
        flags_local, slot_lo, slot_hi, span = engine.get_flags()
        if int(asym.item()):  # every rank merged all lists into a compact node-ordered array: ragged exchange
            flags_all, _ = comm.allgather_ragged(flags_local)
            assert flags_all.numel() == span, (flags_all.numel(), span)
            engine.set_flags(flags_all)
        else:                 # same rank-major padded layout as the rows
            flags_pad = engine.buffer("flags", span, torch.uint8)
            flags_pad[slot_lo:slot_hi] = flags_local
            comm.allgather_inplace(flags_pad, span // world)
            engine.set_flags(flags_pad)
    lap("exchange_survivors")
    e_out_local = engine.emit_edges()
    lap("emit")
    tot = torch.tensor([e_out_local], dtype=torch.int64, device=keys.device)
    comm.all_reduce(tot, "sum")
    return dict(e_pre=engine.total_edges() // 2, e_out_local=e_out_local, e_out=int(tot[0].item()), n_contained=n_contained,
                asymmetric_pairs=n_asym, range=(lo, hi), exchange="rows64")


def _exchange_half(engine, comm, world, rank, per, nloc, n):
    """in-place all-gather of every node's survivor list (4 entries) and survivor count"""
    half_pad = engine.buffer("half", world * per * 4, torch.int64)
    hcnt_pad = engine.buffer("hcnt", world * per, torch.int32)
    engine.export_half(half_pad[rank * per * 4:(rank * per + nloc) * 4], hcnt_pad[rank * per:rank * per + nloc])
    comm.allgather_inplace(half_pad, per * 4)
    comm.allgather_inplace(hcnt_pad, per)
    engine.import_half(half_pad[: n * 4], hcnt_pad[:n])


class HipEngine:
    """adapts BuildGraph (C-ABI, device pointers) to the engine protocol with torch CUDA tensors"""

    def __init__(self, g, device):
        self.g = g
        self.device = device
        self._bufs = {}

    @property
    def num_reads(self):
        return self.g.num_reads

    def buffer(self, name, numel, dtype):
        """persistent device tensor (re-allocated only when it has to grow)"""
        t = self._bufs.get(name)
        if t is None or t.dtype != dtype or t.numel() < numel:
            self._bufs[name] = None
            t = torch.empty(max(int(numel * 1.05) + 16, 16), dtype=dtype, device=self.device)
            self._bufs[name] = t
        return t[:numel]

    def build_index(self):
        self.g.build_index()

    def set_query_range(self, lo, hi):
        self.lo, self.hi = lo, hi
        self.g.set_query_range(lo, hi)

    def probe(self):
        self.g.probe()

    def get_keys(self):
        ptr, n = self.g.contain_keys()
        t = self.buffer("keys", n, torch.int64)
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

    def adjacency_size(self):
        return self.g.adjacency_size()

    def total_edges(self):
        return self.g.adjacency_size()

    def export_adjacency(self, deg_view=None, rows_view=None):
        """fills the given views (local node range / local entries); without arguments returns fresh tensors"""
        total = self.g.adjacency_size()
        if deg_view is None:
            deg_view = torch.empty(max(self.hi - self.lo, 1), dtype=torch.int32, device=self.device)[: self.hi - self.lo]
            rows_view = torch.empty(max(total, 1), dtype=torch.int64, device=self.device)[:total]
        assert rows_view.numel() >= total
        torch.cuda.synchronize(self.device)
        self.g.export_adjacency(deg_view.data_ptr() if deg_view.numel() else self.buffer("dummy", 16, torch.int32).data_ptr(),
                                rows_view.data_ptr() if total else 0)
        return deg_view, rows_view[:total]

    supports_compact = True

    def dropped_hits(self):
        return self.g.dropped_hits()

    def set_global_dropped(self, n_all):
        self.g.set_global_dropped(n_all)

    def export_adjacency32(self, deg_view, rows32_view):
        torch.cuda.synchronize(self.device)
        total = self.g.adjacency_size()
        assert rows32_view.numel() >= total
        self.g.export_adjacency32(deg_view.data_ptr() if deg_view.numel() else self.buffer("dummy", 16, torch.int32).data_ptr(),
                                  rows32_view.data_ptr() if total else 0)

    def adopt_neighbours32(self, deg_pad, rows32_pad, per, mx, world):
        torch.cuda.synchronize(self.device)
        self._adopted = (deg_pad, rows32_pad)  # keep the tensors alive while the context addresses them
        self.g.adopt_neighbours32(deg_pad.data_ptr(), rows32_pad.data_ptr(), per, mx, world)

    def import_adjacency(self, deg_all, rows_all):
        torch.cuda.synchronize(self.device)
        assert deg_all.numel() == self.num_reads
        self.g.import_adjacency(deg_all.data_ptr(), rows_all.data_ptr() if rows_all.numel() else 0, rows_all.numel())

    def adopt_adjacency(self, deg_pad, rows_pad, per, mx, world):
        torch.cuda.synchronize(self.device)
        self._adopted = (deg_pad, rows_pad)  # keep the tensors alive while the context addresses them
        self.g.adopt_adjacency(deg_pad.data_ptr(), rows_pad.data_ptr(), per, mx, world)

    def symmetrize(self, full):
        return self.g.symmetrize(full)

    def merge_extras(self):
        self.g.merge_extras()

    def transitive_mark(self):
        self.g.transitive_mark()

    def n_wide(self):
        """local nodes with more than 4 surviving edges; 'all of them' when the survivor lists are disabled"""
        from .buildgraph import DiscoError

        try:
            return self.g.half_lists()[2]
        except DiscoError:
            return max(self.hi - self.lo, 1)

    def export_half(self, half_view, hcnt_view):
        hp, cp, _ = self.g.half_lists()
        nloc = self.hi - self.lo
        if nloc:
            self.g.memcpy_d2d(half_view.data_ptr(), hp + self.lo * 32, nloc * 32)
            self.g.memcpy_d2d(hcnt_view.data_ptr(), cp + self.lo * 4, nloc * 4)

    def import_half(self, half_all, hcnt_all):
        torch.cuda.synchronize(self.device)
        hp, cp, _ = self.g.half_lists()
        n = self.num_reads
        if n:
            self.g.memcpy_d2d(hp, half_all.data_ptr(), n * 32)
            self.g.memcpy_d2d(cp, hcnt_all.data_ptr(), n * 4)
        self.g.half_complete(True)

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
        assert flags_all.numel() == total
        if total:
            self.g.memcpy_d2d(ptr, flags_all.data_ptr(), total)

    def emit_edges(self):
        return self.g.emit_edges()
