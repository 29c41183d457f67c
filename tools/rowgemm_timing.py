"""Debug: phase stamps (s_memtime = shader cycles; wave 0 of the first 8 workgroups) of the two row-block products a fused encoder layer
still launches: the QKV projection (forward, K = D) and the QKV input gradient (backward, K = 3D).  On the token plan's row count."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, ops, synth
lib = _lib.load()
lib.rd_debug_set_rowgemm_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
T, B, F = 60, int(sys.argv[1]) if len(sys.argv) > 1 else 141, 34
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
    ops.encoder_layer(x, mask, shp, 0, 0.2, 5, pd).backward(dy)
stamps = torch.zeros(8 * 16, dtype=torch.int64, device=dev)
names = ["A load + panel issue + split", "barrier", "export + mma r0", "stage write", "barrier", "epilogue r0", "rounds 1.. + end"]


def show(tag):
    s = stamps.cpu().view(8, 16)
    print(tag)
    for w in range(4):
        d = [int(s[w, i + 1] - s[w, i]) for i in range(7)]
        print("  wg%d" % w, " | ".join("%s %d" % (n, v) for n, v in zip(names, d)), "| total", int(s[w, 7] - s[w, 0]))


lib.rd_debug_set_rowgemm_stamps(stamps.data_ptr())
y = ops.encoder_layer(x, mask, shp, 0, 0.2, 5, pd)
torch.cuda.synchronize()
show("QKV projection, forward (N = 3D = 456, K = D = 152)")
stamps.zero_()
y.backward(dy)
torch.cuda.synchronize()
lib.rd_debug_set_rowgemm_stamps(None)
show("QKV input gradient, backward (N = D = 152, K = 3D = 456; + ds1 residual)")
