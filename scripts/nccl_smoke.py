"""Developer check: the torch.distributed calls bench.py makes for N > 1, on the RCCL backend with a
single rank (all a one-GPU box allows): init with device_id, all_reduce, gather of f64 and u8 tile
buffers into views of one tensor, barrier."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(12, dtype=torch.float64, device=dev)
g = torch.empty((1, 12), dtype=torch.float64, device=dev)
dist.gather(x, list(g.unbind(0)), dst=0)
u = torch.arange(12, dtype=torch.uint8, device=dev); gu = torch.empty((1, 12), dtype=torch.uint8, device=dev)
dist.gather(u, list(gu.unbind(0)), dst=0)
c = torch.tensor([1, 2, 3], dtype=torch.int64, device=dev); dist.all_reduce(c)
t = torch.tensor([1.0, 2.0], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(); torch.cuda.synchronize()
assert torch.equal(g[0], x) and torch.equal(gu[0], u)
dist.destroy_process_group()
print("nccl smoke ok")
