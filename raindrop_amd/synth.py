"""Dataset-shaped configurations, seeded synthetic batches and reproducible parameter fills.

No real data ships with the reference (`/root/reference/.MISSING_LARGE_BLOBS`), so every test,
golden vector and benchmark runs on synthetic inputs of the shapes the training script uses
(`code/Raindrop.py:109-148`).  Everything here is driven by numpy's PCG64 `default_rng`, whose
stream is stable across machines, so the GPU box can regenerate bit-identical inputs and weights
from a seed instead of shipping multi-megabyte fixtures.
"""
import zlib

import numpy as np
import torch

D_OB = 4      # code/Raindrop.py:126
D_PE = 16     # code/models_rd.py:216


def make_config(name):
    """Hyper-parameters exactly as `code/Raindrop.py:109-148` derives them per dataset."""
    table = {
        # name: (d_inp, max_len, d_static, n_classes, static)
        "PAM": (17, 600, 0, 8, False),     # Raindrop.py:121-124,142-144
        "P12": (36, 215, 9, 2, True),      # Raindrop.py:109-112,133-135
        "P19": (34, 60, 6, 2, True),       # Raindrop.py:113-116,136-138
        "SYN256": (256, 512, 6, 2, True),  # BASELINE.json configs[4] (stress shape)
        "TINY": (5, 7, 3, 2, True),        # edge-case shape for fast tests
        "WIDE80": (80, 24, 6, 2, True),    # more sensors than one workgroup's LDS holds as a graph (use_beta operator beyond 64 nodes)
    }
    d_inp, max_len, d_static, n_classes, static = table[name]
    d_model = d_inp * D_OB
    return dict(name=name, d_inp=d_inp, d_model=d_model, nhead=2, nhid=2 * d_model, nlayers=2,
                dropout=0.2, max_len=max_len, d_static=d_static, MAX=100, aggreg="mean",
                n_classes=n_classes, static=static, d_ob=D_OB)


def make_batch(cfg, batch, seed=0, density=None, min_len=2, dtype=torch.float32):
    """Seeded synthetic batch with the layout `model.forward` receives (`code/Raindrop.py:310-317`).

    Returns dict(src[T,B,2F], static[B,d_static] or None, times[T,B], lengths[B] int64, y[B] int64).
    P19-like statistics: per-sample length ~ U[min_len, T]; observation
    indicator Bernoulli(0.9) for the first 8 sensors ("vitals") and Bernoulli(0.06) for the rest
    ("labs"), zero at padded steps; values ~ N(0,1) where observed; times = cumsum(U(0.01,1.01))
    on valid steps and 0 on padding, so `lengths = sum(times > 0)` equals the true length.
    PAM-like (static=False): full length, 40 % density, `times = linspace(0,T,T)/60`
    (first timestamp 0, so `lengths` undercounts by one exactly as `utils_rd.py:248` causes).
    """
    rng = np.random.default_rng(seed)
    T, F = cfg["max_len"], cfg["d_inp"]
    if cfg["static"]:
        length = rng.integers(min_len, T + 1, size=batch)
        p_obs = np.full(F, 0.06)
        p_obs[: min(8, F)] = 0.9
        if density is not None:
            p_obs[:] = density
        valid = (np.arange(T)[:, None] < length[None, :])                       # [T,B]
        obs = (rng.random((T, batch, F)) < p_obs[None, None, :]) & valid[:, :, None]
        vals = rng.standard_normal((T, batch, F)) * obs
        times = np.cumsum(rng.random((T, batch)) + 0.01, axis=0) * valid
        static = rng.standard_normal((batch, cfg["d_static"])).astype(np.float32)
    else:
        p = 0.4 if density is None else density
        obs = rng.random((T, batch, F)) < p
        vals = rng.standard_normal((T, batch, F)) * obs
        times = np.repeat((np.linspace(0, T, T) / 60.0)[:, None], batch, axis=1)
        static = None
    src = np.concatenate([vals, obs.astype(np.float64)], axis=-1).astype(np.float32)
    times = times.astype(np.float32)
    y = np.concatenate([np.zeros(batch // 2), np.ones(batch - batch // 2)]).astype(np.int64) \
        % cfg["n_classes"]
    out = dict(src=torch.from_numpy(src).to(dtype), times=torch.from_numpy(times).to(dtype),
               static=None if static is None else torch.from_numpy(static).to(dtype),
               y=torch.from_numpy(y))
    out["lengths"] = torch.sum(out["times"] > 0, dim=0)                         # Raindrop.py:317
    return out


def make_structure(cfg, kind="ones", seed=0):
    """`global_structure` [F,F]: the shipped all-ones (`code/Raindrop.py:212`) or a sparse,
    non-uniform non-negative matrix (exercises the general edge-softmax path)."""
    F = cfg["d_inp"]
    if kind == "ones":
        return torch.ones(F, F)
    rng = np.random.default_rng(1000 + seed)
    a = rng.random((F, F)) * (rng.random((F, F)) < 0.35)
    return torch.from_numpy(a.astype(np.float32))


def param_values(name, shape, seed=0, scale=1.0):
    """Reproducible values for the parameter called `name`, keyed by (seed, crc32(name)) so they
    do not depend on registration order: U(-b, b) with b = scale/sqrt(fan_in) for matrices
    (fan_in = last dim), U(-0.1, 0.1) for vectors, LayerNorm weights centred on 1."""
    rng = np.random.default_rng([77_000 + seed, zlib.crc32(name.encode())])
    shape = tuple(shape)
    if len(shape) >= 2:
        bound = scale / np.sqrt(shape[-1])
        v = rng.uniform(-bound, bound, size=shape)
    else:
        v = rng.uniform(-0.1, 0.1, size=shape)
        if "norm" in name and name.endswith("weight"):
            v = v + 1.0
    return torch.from_numpy(v.astype(np.float32))


def fill_params_(module, seed=0, scale=1.0):
    """Overwrite EVERY parameter of `module` with `param_values(name, ...)`.  Applied identically
    to the reference model (oracle side) and to ours, so both hold bit-identical weights without
    shipping a state_dict."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            p.copy_(param_values(name, p.shape, seed, scale).to(p.device))
    return module


def live_parameter_names(cfg):
    """Parameters that receive a gradient on the default path (SURVEY.md App. A.6 [probe])."""
    names = ["R_u"]
    if cfg["static"]:
        names += ["emb.weight", "emb.bias"]
    for lyr in ("ob_propagation", "ob_propagation_layer2"):
        names += [lyr + ".lin_value.weight", lyr + ".lin_value.bias"]
    for i in range(cfg["nlayers"]):
        pre = "transformer_encoder.layers.%d." % i
        names += [pre + s for s in (
            "self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
            "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight",
            "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")]
    names += ["mlp_static.0.weight", "mlp_static.0.bias", "mlp_static.2.weight", "mlp_static.2.bias"]
    return names
