import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
x = torch.ones(500000, device=dev)
for _ in range(3):
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        y = x * 2; dist.all_reduce(y, op=dist.ReduceOp.SUM)
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
buf = torch.zeros_like(x)
try:
    with torch.cuda.graph(g):
        buf.copy_(x); buf.mul_(3.0)
        w = dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=True)
        z = x + 1.0                      # work beside the collective
        w.wait()
        out = buf + z
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("CAPTURE OK", float(out[0]), "expected", 3.0 + 2.0)
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    th = time.perf_counter() - t0; torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print("host us/replay %.1f  wall us/replay %.1f" % (th / 200 * 1e6, t1 / 200 * 1e6))
except Exception as e:
    print("CAPTURE FAILED", repr(e)[:300])
dist.destroy_process_group()
