"""Debug: event-timed weight-gradient product (split-K GEMM + fixed-order reduce) for the shapes of the
path, swept over the slab form's rows per workgroup (0 = tiled split-K form)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, ops
lib = _lib.load()
lib.rd_debug_set_wgrad_slabs.argtypes = [ctypes.c_int]
dev = torch.device("cuda")
shapes = [(15360, 152, 272), (15360, 272, 152), (15360, 152, 152), (15360, 456, 152), (8704, 240, 240)]
for want in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 6]:
    lib.rd_debug_set_wgrad_slabs(want)
    line = []
    for (M, N, K) in shapes:
        dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
        dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
        nws = lib.rd_linear_bwd_weight_workspace_bytes(M, N, K)
        ws = torch.empty(max(nws, 256), dtype=torch.uint8, device=dev)
        def run():
            _lib.call("rd_linear_bwd_weight", M, N, K, ops._ptr(dy), N, ops._ptr(x), K, ops._ptr(dW), ops._ptr(db),
                      ops._ptr(ws), ws.numel(), ops._stream())
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        if want > 0 and os.environ.get("RD_WG_STAMPS"):
            lib.rd_debug_set_wgrad_stamps.argtypes = [ctypes.c_void_p]
            stamps = torch.zeros(5 * 128, dtype=torch.int64, device=dev)
            lib.rd_debug_set_wgrad_stamps(stamps.data_ptr()); run(); torch.cuda.synchronize()
            lib.rd_debug_set_wgrad_stamps(None)
            st = stamps.cpu().view(5, 8, 16); st = st[[i for i in range(5) if int(st[i, 0, 0]) != 0][0]]
            names = ["load+convert s0", "barrier", "mma s0 (+issue s1)", "barrier", "rest of slabs", "epilogue"]
            for w in range(3):
                print("   %dx%d wg%d " % (N, K, w) + " ".join("%s=%d" % (n, int(st[w, i + 1] - st[w, i])) for i, n in enumerate(names)),
                      "total", int(st[w, 6] - st[w, 0]))
        ref = dy.double().t() @ x.double()
        err = float((dW.double() - ref).abs().max() / ref.abs().max())
        line.append("%dx%d: %.1f us (ws %.1f MB, err %.1e)" % (N, K, e0.elapsed_time(e1) / 30 * 1e3, nws / 1e6, err))
    print("slabs/wg=%d  " % want + "  ".join(line))
