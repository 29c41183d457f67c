"""Shared test helpers: golden loading, model construction mirrored on both sides."""
import json
import os

import numpy as np
import torch

from raindrop_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_CASES = ["tiny_sparse", "p19_ones", "p19_sparse", "p12_ones", "pam_ones", "p19_b256", "p12_b32", "syn256_b2"]
# the paper's branch (reference with its use_beta literal flipped in memory, tests/golden/make_goldens.py BETA_CASES)
BETA_CASES = ["p19_beta_sparse", "p19_beta_ones", "p12_beta_sparse", "wide80_beta_sparse"]


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(g["meta"]))
    return g, meta


def case_inputs(meta):
    cfg = synth.make_config(meta["cfg"])
    gs = synth.make_structure(cfg, meta["structure"])
    batch = synth.make_batch(cfg, meta["batch"], seed=meta["batch_seed"])
    return cfg, gs, batch


def zero_dropout(model):
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0


def golden_grad(g, name, full):
    """Return (expected strided sample, our tensor sampled the same way)."""
    stride = int(g["gradstride/" + name])
    return g["grad/" + name], full.detach().reshape(-1)[::stride].cpu().numpy()


def build_ours(cfg, gs, device, param_seed, **extra):
    from raindrop_amd.models_rd import Raindrop_v2
    kw = {} if cfg["static"] else {"static": False}
    kw.update(extra)
    m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], cfg["nhead"], cfg["nhid"], cfg["nlayers"],
                    cfg["dropout"], cfg["max_len"], cfg["d_static"], cfg["MAX"], 0.5, cfg["aggreg"],
                    cfg["n_classes"], gs, sensor_wise_mask=False, **kw)
    synth.fill_params_(m, seed=param_seed)
    zero_dropout(m)
    return m.to(device)
