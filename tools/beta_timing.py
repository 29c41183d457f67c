"""The use_beta graph operator alone at the P19 benchmark shape (B = 256 sample graphs, 34 nodes, 1156 edges, 60 steps): forward and
backward launches, HIP events over 20 calls each, rounds 2-5's kernels (RD_BETA_V1=1) against round 6's.   python tools/beta_timing.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import restatement as O2
from raindrop_amd import ops, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n, T, d = 34, 60, 4
K = T * d
dev = "cuda"
rng = np.random.default_rng(0)
ei, ew = O2.build_graph(np.ones((n, n), np.float32))
V = torch.from_numpy(rng.standard_normal((B, n, K)).astype(np.float32)).to(dev).requires_grad_(True)
H = torch.from_numpy(rng.standard_normal((B, n, T * 32)).astype(np.float32)).to(dev).requires_grad_(True)
mw = torch.from_numpy(rng.standard_normal((n, 16)).astype(np.float32)).to(dev).requires_grad_(True)
pt = torch.from_numpy(rng.standard_normal((B, T, 16)).astype(np.float32)).to(dev)
R = torch.from_numpy(rng.standard_normal((B, n, K)).astype(np.float32)).to(dev)
eid, ewd = torch.from_numpy(ei).to(dev), torch.from_numpy(ew).to(dev).reshape(1, -1)
bytes_f = B * 4.0 * (n * K + n * 32 * T + 16 * T + n * K); bytes_b = B * 4.0 * (2 * n * K + 2 * n * 32 * T + 16 * T + 2 * n * K)
for v1 in ("1", "0", "1", "0"):
    os.environ["RD_BETA_V1"] = v1
    tf, tb = [], []
    for it in range(23):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        Y, ei2, al = ops.graph_beta(V, H, mw, pt, eid, ewd, d)
        e[1].record()
        torch.autograd.grad((Y * R).sum(), [V, H, mw])
        e[2].record(); torch.cuda.synchronize()
        if it >= 3:
            tf.append(e[0].elapsed_time(e[1]) * 1e3); tb.append(e[1].elapsed_time(e[2]) * 1e3)
    f, bw = float(np.median(tf)), float(np.median(tb))
    print("%s  fwd %.1f us (%.1f %% of HBM peak)   bwd incl. the two torch kernels of the test's loss %.1f us (%.1f %%)" % (
        "rounds 2-5 kernels" if v1 == "1" else "round 6 kernels   ", f, bytes_f / (f * 1e-6) / 8e12 * 100, bw, bytes_b / (bw * 1e-6) / 8e12 * 100))
