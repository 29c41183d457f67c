"""CPU: the nn.Module surface mirrors the reference's (constructor, parameter names/shapes,
live-parameter set) -- checked against the state_dict surface captured from the reference."""
import json
import os

import pytest
import torch

from raindrop_amd import _lib, synth
from tests.helpers import GOLDEN, build_ours


@pytest.mark.parametrize("cfg_name", ["TINY", "P19", "P12"])
def test_state_dict_surface_matches_reference(cfg_name):
    surf = json.load(open(os.path.join(GOLDEN, "state_dict_surface.json")))[cfg_name]
    cfg = synth.make_config(cfg_name)
    m = build_ours(cfg, synth.make_structure(cfg, "ones"), "cpu", 0)
    ours = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert ours == surf


def test_positional_call_like_training_script():
    # code/Raindrop.py:245-247 constructs the model positionally
    from raindrop_amd.models_rd import Raindrop_v2
    cfg = synth.make_config("TINY")
    gs = torch.ones(cfg["d_inp"], cfg["d_inp"])
    m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"],
                    100, 0.5, "mean", 2, gs, sensor_wise_mask=False)
    assert m.R_u.shape == (1, cfg["d_inp"] * 4)
    assert any(p is m.R_u for p in m.parameters())
    bound = (6.0 / (1 + cfg["d_inp"] * 4)) ** 0.5
    assert float(m.R_u.abs().max()) <= bound + 1e-6          # glorot, code/models_rd.py:276
    assert float(m.emb.weight.abs().max()) <= 1e-10           # code/models_rd.py:272-275


def test_forward_refuses_cpu_tensors():
    cfg = synth.make_config("TINY")
    m = build_ours(cfg, synth.make_structure(cfg, "ones"), "cpu", 0)
    b = synth.make_batch(cfg, 2)
    with pytest.raises(_lib.RaindropHipError):
        m(b["src"], b["static"], b["times"], b["lengths"])


def test_sensor_wise_mask_is_rejected():
    from raindrop_amd.models_rd import Raindrop_v2
    with pytest.raises(_lib.RaindropHipError):
        Raindrop_v2(5, 20, 2, 40, 2, 0.2, 7, 3, 100, 0.5, "mean", 2, torch.ones(5, 5), sensor_wise_mask=True)


def test_synthetic_batch_contract():
    cfg = synth.make_config("P19")
    b = synth.make_batch(cfg, 16, seed=1)
    assert b["src"].shape == (60, 16, 68) and b["times"].shape == (60, 16)
    assert int(b["lengths"].min()) >= 2 and int(b["lengths"].max()) <= 60
    # values are zero wherever the observation indicator is zero, and on padded steps
    assert torch.all(b["src"][:, :, :34][b["src"][:, :, 34:] == 0] == 0)
    b2 = synth.make_batch(cfg, 16, seed=1)
    assert torch.equal(b["src"], b2["src"])                  # reproducible from the seed


def test_legacy_raindrop_surface_matches_reference():
    """The legacy `Raindrop` class (code/models_rd.py:46-191, exported by `from models_rd import *`): same positional constructor,
    same state_dict keys / shapes / order as the reference's (captured in tests/golden/legacy_v1.npz), same init_weights."""
    import numpy as np
    from raindrop_amd import models_rd
    from raindrop_amd.models_rd import Raindrop
    assert "Raindrop" in models_rd.__all__
    g = np.load(os.path.join(GOLDEN, "legacy_v1.npz"), allow_pickle=False)
    meta, surf = json.loads(str(g["meta"])), json.loads(str(g["surface"]))
    m = Raindrop(meta["d_inp"], meta["d_model"], meta["nhead"], meta["nhid"], meta["nlayers"], 0.3, meta["max_len"], meta["d_static"],
                 100, 0.5, "mean", 2, torch.ones(36, 36))
    ours = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert list(ours.items()) == list(surf.items())                 # names, shapes AND registration order
    assert float(m.encoder.weight.abs().max()) <= 1e-10 and float(m.emb.weight.abs().max()) <= 1e-10   # code/models_rd.py:110-113
    with pytest.raises(_lib.RaindropHipError):
        Raindrop(36, 64, 4, 128, 2, 0.3, 215, 9, 100, 0.5, "mean", 2, torch.ones(36, 36))   # upstream's own defaults do not compose


def test_flat_adam_host_state():
    """FlatAdam's host side (no launch): the constants a captured step holds (`hyper`: betas, eps), what lives in the device cell (`cell_hyper`: lr, weight decay), and `load_state_dict` in place -- the moment
    buffers keep their addresses (a captured step stays valid) and the device step state is marked for a re-sync."""
    from raindrop_amd.optim import FlatAdam
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.zeros(8)
    a = FlatAdam(p, lr=1e-4)
    assert a.hyper() == (0.9, 0.999, 1e-8) and a.cell_hyper() == (1e-4, 0.0) and a.step_cell is None and not a._cell_stale
    m_ptr, v_ptr = a.exp_avg.data_ptr(), a.exp_avg_sq.data_ptr()
    sd = dict(a.state_dict(), t=7, lr=5e-5, exp_avg=torch.ones(8), exp_avg_sq=torch.full((8,), 2.0))
    a.load_state_dict(sd)
    assert a.t == 7 and a.cell_hyper()[0] == 5e-5 and a._cell_stale
    assert a.exp_avg.data_ptr() == m_ptr and a.exp_avg_sq.data_ptr() == v_ptr
    assert float(a.exp_avg.sum()) == 8.0 and float(a.exp_avg_sq.sum()) == 16.0
    with pytest.raises(ValueError):
        FlatAdam(torch.nn.Parameter(torch.zeros(2)))
