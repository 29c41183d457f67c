"""Shared test helpers: golden loading, model construction mirrored on both sides."""
import json
import os

import numpy as np
import torch

from raindrop_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODEL_CASES = ["tiny_sparse", "p19_ones", "p19_sparse", "p12_ones", "pam_ones", "p19_b256", "p12_b32", "syn256_b2",
               "p19_x3_sparse"]      # the last: weights at 3x the init scale (make_goldens.py SCALED_CASES; meta["param_scale"])
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


def oracle_params(meta, live=None, requires_grad=False):
    """{state_dict name: tensor} of a golden case for the restatement (oracle/restatement.py), regenerated from the case's seeds."""
    surf = json.load(open(os.path.join(GOLDEN, "state_dict_surface.json")))[meta["cfg"]]
    sc = float(meta.get("param_scale", 1.0))
    return {k: synth.param_values(k, surf[k], meta["param_seed"], sc).requires_grad_(requires_grad and (live is None or k in live))
            for k in sorted(surf) if live is None or k in live}


def gate_unit_masks(meta, rel_eps=3e-5):
    """Which ReLU units of a golden case sit on the fence: {parameter name: bool [units]} -- True where, for some (live) row of
    the batch, the unit's pre-activation lies within rel_eps x rms(pre-activations) of zero, by the fp32 restatement (oracle O2,
    which agrees with the reference to ~1e-6).  A reduced-precision forward (split bf16: ~1e-5) opens / closes exactly those gates
    differently, and ONE flipped gate moves one row of that Linear's weight gradient by |dh| |x| -- 1e-3..1e-2 of the matrix norm
    at a few hundred rows -- while every other row agrees to ~1e-5.  Tests compare the rows of the masked units separately."""
    from oracle import restatement as O2
    cfg, gs, b = case_inputs(meta)
    p = oracle_params(meta)
    with torch.no_grad():
        _, _, inter = O2.raindrop_v2_forward(p, cfg, b["src"], b["static"], b["times"], b["lengths"], gs, return_intermediates=True)
    T, B = b["src"].shape[0], b["src"].shape[1]
    live_tb = (torch.arange(T)[:, None] < b["lengths"][None, :])                 # [T, B] live (sample, step) rows
    out = {}
    for name, z in inter["gates"].items():
        rms = float(z.pow(2).mean().sqrt()) + 1e-30
        near = z.abs() < rel_eps * rms
        if name.startswith("ob_propagation"):                                    # [B, F, K]: every sensor row of every sample
            if "layer2" in name:                                                 # layer 2's units at a sample's padded steps feed nothing
                K = z.shape[2]
                step_live = (torch.arange(K)[None, :] // cfg["d_ob"]) < b["lengths"][:, None]     # [B, K]
                near = near & step_live[:, None, :]
            units = near.reshape(-1, near.shape[-1]).any(0)
        elif name.startswith("transformer_encoder"):                             # [T, B, nhid]: live tokens only
            units = (near & live_tb[:, :, None]).reshape(-1, near.shape[-1]).any(0)
        else:                                                                    # mlp_static.0: [B, dh]
            units = near.any(0)
        out[name + ".weight"] = units.numpy()
        out[name + ".bias"] = units.numpy()
    return out


def build_ours(cfg, gs, device, param_seed, param_scale=1.0, **extra):
    from raindrop_amd.models_rd import Raindrop_v2
    kw = {} if cfg["static"] else {"static": False}
    kw.update(extra)
    m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], cfg["nhead"], cfg["nhid"], cfg["nlayers"],
                    cfg["dropout"], cfg["max_len"], cfg["d_static"], cfg["MAX"], 0.5, cfg["aggreg"],
                    cfg["n_classes"], gs, sensor_wise_mask=False, **kw)
    synth.fill_params_(m, seed=param_seed, scale=param_scale)
    zero_dropout(m)
    return m.to(device)
