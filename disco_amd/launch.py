"""control plane of a multi-rank run (one process per GPU): rendezvous, the broadcast of the RCCL unique id, barrier, max over
ranks. Replaces what mpirun + MPI_Init_thread / MPI_Comm_rank / MPI_Bcast give the reference's multi-process binaries
(MPI/main.cpp:29-37). Only HOST values travel here (torch.distributed, backend gloo); every device buffer is exchanged by RCCL
inside libdisco_hip.so (disco_comm_init, disco_dist_run_graph)."""
from __future__ import annotations

import os


def rank_env():
    """(rank, world, local_rank) from the launcher's environment (torchrun / torch.distributed.run)"""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def gpu_count() -> int:
    """GPUs of this machine WITHOUT touching the HIP runtime: the KFD topology in sysfs lists every agent, a GPU is a node with
    SIMDs. A launcher that only starts child ranks must not open /dev/kfd itself (torch.cuda.device_count() falls back to
    hipGetDeviceCount — hipInit — on ROCm builds without amdsmi). HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES narrow the count
    the way they narrow what the ranks will see."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    n = 0
    try:
        for node in os.listdir(root):
            try:
                with open(os.path.join(root, node, "properties")) as fh:
                    props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def owner_range(n_total: int, rank: int, world: int):
    """[lo, hi) of the reads / graph nodes a rank owns: ceil(n / world) rounded up to a multiple of 64 per rank — the twin of
    disco_dist_range (include/disco_hip.h), checked against it by tests/test_gpu_dist.py"""
    per = (n_total + world - 1) // world
    per = max((per + 63) // 64 * 64, 64)
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def bucket_range(log2_buckets: int, rank: int, world: int):
    """[lo, hi) of the index buckets a rank builds: the buckets b with (b * world) >> log2_buckets == rank"""
    t = 1 << log2_buckets
    return (rank * t + world - 1) // world, ((rank + 1) * t + world - 1) // world


class ControlPlane:
    def __init__(self, backend: str = "gloo"):
        import torch.distributed as dist

        self.dist = dist
        self.rank, self.world, self.local_rank = rank_env()
        self.active = False
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29511"), RANK=str(self.rank),
                              WORLD_SIZE=str(self.world))
        if not dist.is_initialized():
            dist.init_process_group(backend)
        self.active = True

    def broadcast_unique_id(self, make_id) -> bytes:
        """rank 0 calls make_id() (disco_comm_unique_id); every rank returns the same bytes"""
        box = [make_id() if self.rank == 0 else None]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def barrier(self):
        self.dist.barrier()

    def max_over_ranks(self, x: float) -> float:
        import torch

        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.active and self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()
        self.active = False
