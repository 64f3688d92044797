"""RCCL sanity check for the sharded flow: in-place all_gather_into_tensor (input = slice of the output), int64 MIN/MAX/SUM
all-reduce, a 6.4 GB collective. One rank suffices to see that torch / RCCL accept the calls."""
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
for dt in (torch.int64, torch.int32, torch.uint8):
    buf = torch.arange(1 << 20, device="cuda").to(dt)
    ref = buf.clone()
    dist.all_gather_into_tensor(buf[: 1 << 20], buf[0: 1 << 20])
    torch.cuda.synchronize()
    print(dt, "in-place all_gather_into_tensor ok:", bool((buf == ref).all()))
t = torch.tensor([5, 7], dtype=torch.int64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MIN); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.all_reduce(t, op=dist.ReduceOp.SUM)
print("all_reduce int64 min/max/sum ok", t.tolist())
big = torch.zeros(3 << 28, dtype=torch.int64, device="cuda")  # 6.4 GB: counts beyond 2^31 bytes
dist.all_gather_into_tensor(big, big)
torch.cuda.synchronize()
print("6.4 GB in-place all-gather ok")
dist.destroy_process_group()
