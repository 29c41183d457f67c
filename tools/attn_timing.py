"""Debug: phase stamps (clock64) of the single-tile bf16 attention kernels inside a real encoder layer forward + backward."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, ops, synth
lib = _lib.load()
lib.rd_debug_set_attn_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
T, B, F = 60, 256, 34
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
stamps = torch.zeros(8 * 16, dtype=torch.int64, device=dev)
names = ["issue loads", "wait+split+LDS", "barrier", "S (+dP)", "softmax/P^T", "barrier", "PV | dQ dK dV", "stores"]
lib.rd_debug_set_attn_stamps(stamps.data_ptr())
y = ops.encoder_layer(x, mask, shp, 0, 0.2, 5, pd)
torch.cuda.synchronize()
s = stamps.cpu().view(8, 16).clone()
print("forward  (cycles @100 MHz clock64 -> x24 for 2.4 GHz shader cycles)")
for w in range(4):
    print("wg%d" % w, " ".join("%s=%d" % (n, int(s[w, i + 1] - s[w, i])) for i, n in enumerate(names[:7])), "total", int(s[w, 7] - s[w, 0]))
stamps.zero_()
y.backward(dy)
torch.cuda.synchronize(); lib.rd_debug_set_attn_stamps(None)
s = stamps.cpu().view(8, 16)
print("backward")
for w in range(4):
    print("wg%d" % w, " ".join("%s=%d" % (n, int(s[w, i + 1] - s[w, i])) for i, n in enumerate(names[:7])), "total", int(s[w, 7] - s[w, 0]))
