"""Gradient parity against the O1 fixtures with the gate-flip rows separated (tests/helpers.gate_unit_masks): per case and
parameter, relative L2 / max-norm error over all sampled entries and over the entries of units that are NOT on a ReLU fence.
    python tools/grad_gate_diag.py [case ...]            (GPU; prints one line per parameter whose error exceeds 2e-4)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raindrop_amd import _lib
from tests.helpers import MODEL_CASES, build_ours, case_inputs, gate_unit_masks, golden_grad, load_golden

def rel2(a, b): return float(np.linalg.norm((a - b).astype(np.float64)) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))
def relm(a, b): return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))

cases = sys.argv[1:] or [c for c in MODEL_CASES if c not in ("pam_ones", "syn256_b2")]
for mode in (1, 0):
    _lib.call("rd_set_precision", mode)
    for name in cases:
        g, meta = load_golden(name)
        cfg, gs, batch = case_inputs(meta)
        m = build_ours(cfg, gs, "cuda", meta["param_seed"], float(meta.get("param_scale", 1.0))).train()
        m.graph_step = False
        dv = {k: (None if v is None else v.cuda()) for k, v in batch.items()}
        logits, _, _ = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
        torch.nn.functional.cross_entropy(logits, dv["y"]).backward()
        masks = gate_unit_masks(meta)
        params = dict(m.named_parameters())
        worst = (0, 0, "")
        print("== %s %s  logits %.2e" % ("bf16x3" if mode else "fp32", name, np.abs(logits.detach().cpu().numpy() - g["logits"]).max()))
        for n in [str(x) for x in g["live"]]:
            exp, got = golden_grad(g, n, params[n].grad)
            st = int(g["gradstride/" + n])
            r2, rm = rel2(got, exp), relm(got, exp)
            line = "   %-58s all: l2 %.2e max %.2e" % (n, r2, rm)
            if n in masks:
                cols = params[n].shape[1] if params[n].dim() == 2 else 1
                unit = (np.arange(exp.size) * st) // cols
                keep = ~masks[n][unit]
                if keep.any():
                    line += "   clear units (%d of %d fenced): l2 %.2e max %.2e" % (int(masks[n].sum()), masks[n].size, rel2(got[keep], exp[keep]), relm(got[keep], exp[keep]))
            if r2 > 2e-4 or os.environ.get("ALL"):
                print(line)
