"""Debug: per-phase cycle totals (clock64, summed over the live key tiles) of the multi-tile split-bf16 attention forward at the P12 shape."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, synth
lib = _lib.load()
lib.rd_debug_set_attn_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
T, B, F, nhead = 215, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 36, 2
D = F * 4 + 16
cfg = synth.make_config("P12"); L = synth.make_batch(cfg, max(B, 4), seed=100)["lengths"][:B]
mask = (torch.arange(T)[None, :] >= L[:, None]).to(dev)
qkv = torch.randn(T, B, 3 * D, device=dev)
out = torch.zeros(T, B, D, device=dev); lse = torch.zeros(B, nhead, T, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
shp = _lib.shape(B, T, F, 4, nhead=nhead, nhid=2 * F * 4)
for _ in range(3):
    _lib.call("rd_attention_fwd", ctypes.byref(shp), 0, P(qkv), P(mask), 0.2, 7, P(out), P(lse), None)
stamps = torch.zeros(8 * 16, dtype=torch.int64, device=dev)
lib.rd_debug_set_attn_stamps(stamps.data_ptr())
_lib.call("rd_attention_fwd", ctypes.byref(shp), 0, P(qkv), P(mask), 0.2, 7, P(out), P(lse), None)
torch.cuda.synchronize(); lib.rd_debug_set_attn_stamps(None)
s = stamps.cpu().view(8, 16)
names = ["prologue", "issue+bar1", "wait/split/store+bar2", "S", "softmax+P^T+bar3", "PV", "epilogue", "live tiles"]
for w in range(min(8, B * nhead)):
    print("bh%d len=%d " % (w, int(L[w // nhead])) + " ".join("%s=%d" % (n, int(s[w, i])) for i, n in enumerate(names)))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    _lib.call("rd_attention_fwd", ctypes.byref(shp), 0, P(qkv), P(mask), 0.2, 7, P(out), P(lse), None)
ev[1].record(); torch.cuda.synchronize()
print("fwd us", ev[0].elapsed_time(ev[1]) * 1e3 / 20)
