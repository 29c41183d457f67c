"""How far does Adam amplify a small gradient perturbation on this model?  (CPU; the fp32 restatement, oracle O2.)
The 20 reference steps of tests/golden/p19_traj20.npz (lr 1e-3, dropout 0) are run exactly and with Gaussian noise of a given RELATIVE
L2 size added to every gradient before the optimizer; printed: the drift of the loss trajectory and of the held-out logits under the
trained weights.  Context: tests/test_trajectory_gpu.py (a free-running split-bf16 replay drifts by 7e-3 / 0.19).
    python tools/adam_noise_drift.py > profiles/r06_adam_noise_drift.txt"""
import sys, json; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import restatement as O2
from raindrop_amd import synth
from tests.helpers import load_golden, oracle_params
torch.set_num_threads(8)
g, meta = load_golden("p19_traj20")
cfg = synth.make_config(meta["cfg"]); gs = synth.make_structure(cfg, meta["structure"])
batches = [synth.make_batch(cfg, meta["batch"], seed=meta["batch_seed0"] + i) for i in range(meta["steps"])]
held = synth.make_batch(cfg, meta["batch"], seed=meta["held_out_seed"])
live = [str(x) for x in g["live"]]
def run(rel, seed=0, lr=meta["lr"]):
    p = oracle_params(meta)
    for n in live: p[n].requires_grad_(True)
    opt = torch.optim.Adam([p[n] for n in live], lr=lr)
    gen = torch.Generator().manual_seed(seed); L = []
    for b in batches:
        logits, loss, grads = O2.step_fwd_bwd(p, cfg, b, gs)
        for n in live:
            gr = grads[n]
            if rel: gr = gr + torch.randn(gr.shape, generator=gen) * (rel * gr.norm() / np.sqrt(gr.numel()))
            p[n].grad = gr
        opt.step(); L.append(float(loss))
    with torch.no_grad():
        hl, _ = O2.raindrop_v2_forward({k: v.detach() for k, v in p.items()}, cfg, held["src"], held["static"], held["times"], held["lengths"], gs)
    return np.array(L), hl.numpy()
L0, h0 = run(0.0)
print("exact vs golden: loss %.2e held %.2e" % (np.abs(L0 - g["losses"]).max(), np.abs(h0 - g["held_logits"]).max()))
for rel in (1e-6, 1e-5, 1e-4, 1e-3, 3e-3):
    L, h = run(rel)
    print("rel L2 noise %.0e: loss drift max %.2e, first3 %.2e, held logits %.3f" % (rel, np.abs(L - L0).max(), np.abs(L - L0)[:3].max(), np.abs(h - h0).max()))
for lr in (1e-4,):
    La, ha = run(0.0, lr=lr); Lb, hb = run(1e-3, lr=lr)
    print("lr %.0e, noise 1e-3: loss drift %.2e held %.3f" % (lr, np.abs(La - Lb).max(), np.abs(ha - hb).max()))
