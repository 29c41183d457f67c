"""Debug: wall-clock (100 MHz) and shader-clock stamps at the start and end of EVERY workgroup of k_enc_post_fwd inside a graph-less
TrainStep-like forward: shows launch skew, per-workgroup duration and the effective shader clock."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, ops, synth
lib = _lib.load()
lib.rd_debug_set_encfuse_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
T, B, F = 60, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 34
D, nhid = F * 4 + 16, 2 * F * 4
x = torch.randn(T, B, D, device=dev)
mask = torch.zeros(B, T, dtype=torch.bool, device=dev)
shapes = {"self_attn.in_proj_weight": (3 * D, D), "self_attn.in_proj_bias": (3 * D,), "self_attn.out_proj.weight": (D, D),
          "self_attn.out_proj.bias": (D,), "linear1.weight": (nhid, D), "linear1.bias": (nhid,), "linear2.weight": (D, nhid),
          "linear2.bias": (D,), "norm1.weight": (D,), "norm1.bias": (D,), "norm2.weight": (D,), "norm2.bias": (D,)}
pd = [synth.param_values(n, shapes[n], 1).to(dev) for n in ops.ENC_PARAM_NAMES]
shp = _lib.shape(B, T, F, 4, nhead=2, nhid=nhid)
with torch.no_grad():
    for _ in range(3):
        ops.encoder_layer(x, mask, shp, 0, 0.2, 5, pd)
    stamps = torch.zeros(256 + 2048, dtype=torch.int64, device=dev)
    lib.rd_debug_set_encfuse_stamps(stamps.data_ptr())
    ops.encoder_layer(x, mask, shp, 0, 0.2, 5, pd)
    torch.cuda.synchronize()
    lib.rd_debug_set_encfuse_stamps(None)
s = stamps.cpu()
n = T * B // 32
st = s[256:256 + 2 * n].view(n, 2); en = s[256 + 1024:256 + 1024 + 2 * n].view(n, 2)
w0 = int(st[:, 0].min())
dur_w = (en[:, 0] - st[:, 0]).double() * 10e-3     # us (100 MHz ticks)
dur_c = (en[:, 1] - st[:, 1]).double()
print("workgroups", n, "first start -> last end: %.1f us" % ((int(en[:, 0].max()) - w0) * 10e-3))
print("start skew: %.1f us;  per-WG duration us: min %.1f median %.1f max %.1f" % ((int(st[:, 0].max()) - w0) * 10e-3, dur_w.min(), dur_w.median(), dur_w.max()))
print("per-WG shader cycles: min %d median %d max %d -> effective clock %.2f GHz" % (dur_c.min(), dur_c.median(), dur_c.max(), float((dur_c / dur_w).median()) / 1e3))
