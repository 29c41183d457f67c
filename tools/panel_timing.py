"""Debug: clock64 stamps of the panel GEMM inside rd_msgpass_fwd at a dataset's shape (8 workgroups spread over the grid).
python tools/panel_timing.py [P12|PAM] [B]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, synth
lib = _lib.load()
lib.rd_debug_set_gemm_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
cfg = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else "P12"); B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
F, T, d = cfg["d_inp"], cfg["max_len"], 4
K = T * d
shp = _lib.shape(B, T, F, d, nhead=2, nhid=2 * F * d)
src = torch.randn(T, B, 2 * F, device=dev); Ru = torch.randn(F, d, device=dev)
W1 = torch.randn(K, K, device=dev) / K ** 0.5; W2 = torch.randn(K, K, device=dev) / K ** 0.5
b1 = torch.randn(K, device=dev); b2 = torch.randn(K, device=dev); ssum = torch.rand(F, device=dev)
ldz = F * d + 16
z = torch.zeros(T, B, ldz, device=dev)
lib.rd_msgpass_saved_bytes.restype = ctypes.c_size_t
nb = lib.rd_msgpass_saved_bytes(ctypes.byref(shp))
saved = torch.zeros(nb, dtype=torch.uint8, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
run = lambda: _lib.call("rd_msgpass_fwd", ctypes.byref(shp), P(src), P(Ru), P(W1), P(b1), P(W2), P(b2), P(ssum), 0.2, 5, P(z), ldz, P(saved), nb, None)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("rd_msgpass_fwd M=%d K=%d: %.1f us per call" % (B * F, K, e0.elapsed_time(e1) * 100))
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
lib.rd_debug_set_gemm_stamps(stamps.data_ptr()); run(); torch.cuda.synchronize(); lib.rd_debug_set_gemm_stamps(None)
s = stamps.cpu().view(8, 8)
t0 = int(s[:, 0][s[:, 0] > 0].min())
for w in range(8):
    if s[w, 0] == 0: continue
    print("  wg sample %d: start@%7d  prologue +%6d  loop +%6d  epilogue +%6d" % (w, int(s[w, 0]) - t0, int(s[w, 1] - s[w, 0]), int(s[w, 2] - s[w, 1]), int(s[w, 3] - s[w, 2])))
