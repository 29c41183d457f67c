"""Debug: clock64 stamps inside the bf16x3 GEMM (start / first tile staged / k-loop done / epilogue done)
for 8 sample workgroups, plus event-timed kernel duration, for the shapes of the path."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, ops
lib = _lib.load()
lib.rd_debug_set_gemm_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
for (M, N, K) in [(15360, 456, 152), (15360, 152, 152), (15360, 272, 152), (15360, 152, 272), (8704, 240, 240)]:
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    def run():
        _lib.call("rd_linear_fwd", M, N, K, ops._ptr(x), K, ops._ptr(W), ops._ptr(b), ops._ptr(y), N, 0, ops._stream())
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    stamps = torch.zeros(64, dtype=torch.int64, device=dev)
    lib.rd_debug_set_gemm_stamps(stamps.data_ptr())
    run(); torch.cuda.synchronize()
    lib.rd_debug_set_gemm_stamps(None)
    s = stamps.cpu().view(8, 8)
    t0 = int(s[:, 0][s[:, 0] > 0].min())
    print("NT M=%d N=%d K=%d  avg %.1f us (back-to-back)  bytes %.1f MB" % (M, N, K, e0.elapsed_time(e1) / 20 * 1e3, (M * K + M * N) * 4 / 1e6))
    for w in range(8):
        if s[w, 0] == 0: continue
        print("   wg(y=%4d): start@%6d  first-tile +%5d  kloop +%5d  epilogue +%5d" % (
            w * 64, int(s[w, 0]) - t0, int(s[w, 1] - s[w, 0]), int(s[w, 2] - s[w, 1]), int(s[w, 3] - s[w, 2])))
