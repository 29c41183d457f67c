"""Debug: phase stamps (clock64, all 16 waves of workgroup 0) of the fused row-local encoder kernels (rd_encfuse.hip) inside a real
encoder layer forward + backward.  Usage: encfuse_timing.py"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, ops, synth
lib = _lib.load()
lib.rd_debug_set_encfuse_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
T, B, F = 60, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 34
D, nhid = F * 4 + 16, 2 * F * 4
x = torch.randn(T, B, D, device=dev, requires_grad=True)
mask = torch.zeros(B, T, dtype=torch.bool, device=dev)
shapes = {"self_attn.in_proj_weight": (3 * D, D), "self_attn.in_proj_bias": (3 * D,), "self_attn.out_proj.weight": (D, D),
          "self_attn.out_proj.bias": (D,), "linear1.weight": (nhid, D), "linear1.bias": (nhid,), "linear2.weight": (D, nhid),
          "linear2.bias": (D,), "norm1.weight": (D,), "norm1.bias": (D,), "norm2.weight": (D,), "norm2.bias": (D,)}
pd = [synth.param_values(n, shapes[n], 1).to(dev).requires_grad_(True) for n in ops.ENC_PARAM_NAMES]
shp = _lib.shape(B, T, F, 4, nhead=2, nhid=nhid)
dy = torch.randn(T, B, D, device=dev)
for _ in range(3):
    y = ops.encoder_layer(x, mask, shp, 0, 0.2, 5, pd); y.backward(dy)
stamps = torch.zeros(8192, dtype=torch.int64, device=dev)      # forward chain [0, 4096), backward chain [4096, 8192)


def show(tag, n, off=0):
    s = stamps.cpu()[off:off + 256].view(16, 16)
    t0 = int(s[:, 0].min())
    print(tag, "cycles since the first wave started: min .. max over the 16 waves")
    for i in range(n):
        col = s[:, i] - t0
        print("  %2d  %7d .. %7d" % (i, int(col.min()), int(col.max())))


lib.rd_debug_set_encfuse_stamps(stamps.data_ptr())
y = ops.encoder_layer(x, mask, shp, 0, 0.2, 5, pd)
torch.cuda.synchronize()
show("post_fwd: 0 start | 1 rows split | 2 barrier | 3 out_proj staged | 4 barrier | 5 LN1 done | 6 barrier | 7 linear1 + h epilogue done | "
     "8 barrier | 9 linear2 staged | 10 barrier | 11 LN2 done", 12)
stamps.zero_()
y.backward(dy)
torch.cuda.synchronize()
lib.rd_debug_set_encfuse_stamps(None)
show("pre_bwd: 0 start | 1 barrier | 2 LN2' done | 3 barrier | 4 du product + gate done | 5 barrier | 6 dx1 staged | 7 barrier | 8 LN1' done | "
     "9 barrier | 10 d attn stored", 11, 4096)
