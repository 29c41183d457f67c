"""Debug: fused K1 against the generic path in the same arithmetic, every gradient's error printed (tests stop at the first).
   python tools/k1_debug.py [F T B]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from raindrop_amd import _lib, ops, synth
DEV = torch.device("cuda")
F, T, B = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (34, 60, 9)
d, K = 4, T * 4
rng = np.random.default_rng(F * 1000 + T * 10 + B)
cfgF = dict(d_inp=F, max_len=T, static=True, d_static=3, n_classes=2)
b = synth.make_batch(cfgF, B, seed=F + B, density=0.5)
gs = torch.ones(F, F)
names = ["R_u", "W1", "b1", "W2", "b2"]
shapes = [(1, F * d), (K, K), (K,), (K, K), (K,)]
p = {n: synth.param_values("k1." + n, s, seed=3) for n, s in zip(names, shapes)}
p["R_u"] = p["R_u"] * 3.0
dz = torch.from_numpy(rng.standard_normal((T, B, F * d + 16)).astype(np.float32)).to(DEV)
adj, _, _ = ops.graph_build(gs.to(DEV))
_, ssum = ops.edge_softmax_dense(adj)
shp = _lib.shape(B, T, F, d)
out = {}
for mode in ("1", "0", "1"):
    os.environ["RD_K1_FUSED"] = mode
    pd = {n: t.detach().to(DEV).requires_grad_(True) for n, t in p.items()}
    z, mask = ops.sensor_stage(b["src"].to(DEV), b["times"].to(DEV), b["lengths"].to(DEV), ops.timescales(T).to(DEV),
                               ssum, pd["R_u"], pd["W1"], pd["b1"], pd["W2"], pd["b2"], shp)
    g = torch.autograd.grad(z, [pd[n] for n in names], dz)
    torch.cuda.synchronize()
    if mode in out:
        print("fused twice: identical gradients:", [bool(np.array_equal(x.cpu().numpy(), y)) for x, y in zip(g, out[mode][1])])
    out[mode] = (z.detach().cpu().numpy(), [x.cpu().numpy() for x in g])
zf, gf = out["1"]; zg, gg = out["0"]
print("z max err / max", np.abs(zf - zg).max() / np.abs(zg).max())
for n, a, r in zip(names, gf, gg):
    scale = np.abs(r).max() + 1e-30
    print("%-4s max err / max %.3e   rel L2 %.3e   bad %.4f" % (n, np.abs(a - r).max() / scale, np.linalg.norm(a - r) / np.linalg.norm(r),
                                                              (np.abs(a - r) > 2e-5 * scale).mean()))
