"""Debug: achievable HBM bandwidth of plain torch copies / reads on this box (sanity figure beside the kernels' rooflines)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda")
def timeit(fn, n=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M, N, K = 15360, 456, 152
x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); y = torch.empty(M, N, device=dev)
y2 = torch.empty_like(y)
print("copy 28MB->28MB   us", timeit(lambda: y2.copy_(y)), " => GB/s", 2 * y.numel() * 4 / timeit(lambda: y2.copy_(y)) / 1e3)
print("fill 28MB         us", timeit(lambda: y.fill_(1.0)))
print("torch matmul      us", timeit(lambda: torch.matmul(x, W.t(), out=y)))
big = torch.empty(256 * 1024 * 1024 // 4, device=dev); big2 = torch.empty_like(big)
t = timeit(lambda: big2.copy_(big), 10)
print("copy 256MB us", t, "GB/s", 2 * big.numel() * 4 / t / 1e3)
from raindrop_amd import _lib, ops
b = torch.randn(N, device=dev)
def mine():
    _lib.call("rd_linear_fwd", M, N, K, ops._ptr(x), K, ops._ptr(W), ops._ptr(b), ops._ptr(y), N, 0, ops._stream())
print("rd_linear_fwd bf16x3 us", timeit(mine))
_lib.call("rd_set_precision", 0)
print("rd_linear_fwd fp32   us", timeit(mine))
_lib.call("rd_set_precision", 1)
# smaller N
for n2 in (64, 152):
    W2 = torch.randn(n2, K, device=dev); y3 = torch.empty(M, n2, device=dev); b2 = torch.randn(n2, device=dev)
    print("rd_linear N=%d us" % n2, timeit(lambda: _lib.call("rd_linear_fwd", M, n2, K, ops._ptr(x), K, ops._ptr(W2), ops._ptr(b2), ops._ptr(y3), n2, 0, ops._stream())))
