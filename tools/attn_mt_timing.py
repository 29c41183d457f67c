"""Debug: time the multi-tile attention core (rd_attention_fwd / rd_attention_bwd) at the P12 shape.   python tools/attn_mt_timing.py [B]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, synth
dev = torch.device("cuda")
T, B, F, nhead = int(sys.argv[2]) if len(sys.argv) > 2 else 215, int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[3]) if len(sys.argv) > 3 else 36, 2
D = F * 4 + 16
cfg = synth.make_config("P12" if T > 64 else "P19"); L = synth.make_batch(cfg, max(B, 4), seed=100)["lengths"][:B]
mask = (torch.arange(T)[None, :] >= L[:, None]).to(dev)
qkv = torch.randn(T, B, 3 * D, device=dev); dout = torch.randn(T, B, D, device=dev)
out = torch.zeros(T, B, D, device=dev); lse = torch.zeros(B, nhead, T, device=dev)
dqkv = torch.zeros(T, B, 3 * D, device=dev); ws = torch.zeros(B, nhead, T, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
shp = _lib.shape(B, T, F, 4, nhead=nhead, nhid=2 * F * 4)
fwd = lambda: _lib.call("rd_attention_fwd", ctypes.byref(shp), 0, P(qkv), P(mask), 0.2, 7, P(out), P(lse), None)
bwd = lambda: _lib.call("rd_attention_bwd", ctypes.byref(shp), 0, P(qkv), P(mask), 0.2, 7, P(out), P(lse), P(dout), P(dqkv), P(ws), None)
for name, f in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(3): f()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20): f()
    ev[1].record(); torch.cuda.synchronize()
    print(name, "us %.1f" % (ev[0].elapsed_time(ev[1]) * 1e3 / 20))
