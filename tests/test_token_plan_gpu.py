"""GPU: the token plan (include/raindrop_hip.h "token plan", raindrop_amd/csrc/rd_plan.h) -- the padding mask of
code/models_rd.py:298-299 applied as a layout.  INT work (ranks, offsets, counts) bit-exact against numpy; the step on the
compact layout against the same step on the padded layout: same loss, logits and every gradient up to the order of the fp32
sums over tokens."""
import ctypes

import numpy as np
import pytest
import torch

from raindrop_amd import _lib, dp, synth
from tests.helpers import build_ours

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _plan_ref(lengths, T):
    """numpy statement of rd_plan.h: stable descending sort by clamped length."""
    L = np.clip(np.asarray(lengths, dtype=np.int64), 0, T)
    B = len(L)
    order = np.array(sorted(range(B), key=lambda b: (-L[b], b)), dtype=np.int64)
    rank = np.empty(B, dtype=np.int64); rank[order] = np.arange(B)
    lenr = L[order]
    off = np.concatenate([[0], np.cumsum(lenr)])
    cnt = np.array([(L > t).sum() for t in range(T + 1)])
    return dict(off=off, rank=rank, order=order, lenr=lenr, cnt=cnt, mlive=int(off[-1]))


@pytest.mark.parametrize("B,T,seed", [(256, 60, 0), (1, 60, 1), (37, 60, 2), (5, 16, 3), (1000, 60, 4)])
def test_token_plan_bit_exact(B, T, seed):
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    lengths = rng.integers(0, T + 3, size=B)                 # includes 0 and values beyond T (clamped)
    shp = _lib.shape(B, T, 34, 4)
    n = lib.rd_token_plan_bytes(ctypes.byref(shp)) // 4
    plan = torch.full((n,), -7, dtype=torch.int32, device=DEV)
    cell = torch.tensor([41], dtype=torch.int64, device=DEV)
    ld = torch.from_numpy(lengths).to(DEV)
    _lib.call("rd_token_plan", ctypes.byref(shp), ctypes.c_void_p(ld.data_ptr()), ctypes.c_void_p(plan.data_ptr()),
              ctypes.c_void_p(cell.data_ptr()), 3, None)
    torch.cuda.synchronize()
    p = plan.cpu().numpy()
    ref = _plan_ref(lengths, T)
    assert int(cell) == 44
    assert p[0] == ref["mlive"] and p[1] == (ref["mlive"] + 31) // 32 and p[2] == B and p[3] == T and p[4] == 0
    o = 8
    assert np.array_equal(p[o:o + B + 1], ref["off"]); o += B + 1
    assert np.array_equal(p[o:o + B], ref["rank"]); o += B
    assert np.array_equal(p[o:o + B], ref["order"]); o += B
    assert np.array_equal(p[o:o + B], ref["lenr"]); o += B
    assert np.array_equal(p[o:o + T + 1], ref["cnt"])


def _run_step(cfg, gs, batch, token_plan, use_graph, p_drop=0.0, steps=1, seed_params=7):
    from raindrop_amd.step import TrainStep
    dv = {k: (None if v is None else v.to(DEV)) for k, v in batch.items()}
    m = build_ours(cfg, gs, DEV, seed_params).train()
    m.dropout.p = p_drop
    live = synth.live_parameter_names(cfg)
    named = dict(m.named_parameters())
    flat = dp.FlatGradAllReduce([(n, named[n]) for n in live])
    step = TrainStep(m, flat, dv, use_graph=use_graph, token_plan=token_plan, autotune=False)
    assert (step.plan is not None) == bool(token_plan)
    losses = []
    for _ in range(steps):
        losses.append(float(step.run()))
    torch.cuda.synchronize()
    grads = {n: named[n].grad.detach().cpu().numpy().copy() for n in live}
    return losses, step.logits.cpu().numpy().copy(), grads, step


CASES = ["random", "full_length", "min_length", "first_time_zero", "one_long"]


def _case_batch(cfg, B, case, seed):
    batch = synth.make_batch(cfg, B, seed=seed)
    T = cfg["max_len"]
    if case == "full_length":
        batch["times"] = torch.cumsum(torch.rand(T, B) + 0.01, 0)
    if case == "min_length":
        batch["times"][1:] = 0
        batch["src"][1:] = 0
    if case == "first_time_zero":
        # code/Raindrop.py:317 counts steps with time > 0: a first time stamp of 0 (always the case for PAM, utils_rd.py:248) makes
        # `lengths` one short, so the LAST observed step is masked as a key but still feeds the sensor graph (src != 0 there)
        batch["times"][0] = 0
    if case == "one_long":
        batch["times"][:, 1:] = 0
        batch["times"][1, 1:] = 0.5
        batch["src"][2:, 1:] = 0
    batch["lengths"] = torch.sum(batch["times"] > 0, dim=0)
    if case == "min_length":
        assert int(batch["lengths"].max()) == 1
    return batch


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("B,use_graph", [(37, False), (256, True)])
def test_step_on_token_plan_matches_padded_step(case, B, use_graph):
    """Same batch, same parameters, dropout off: TrainStep on the compact layout (plan) against TrainStep on the padded layout.
    The logits agree to rounding of identical per-sample arithmetic (the live rows see the same values; only which zero
    products are skipped differs), the gradients to the order of the sums over tokens: 2e-5 of each tensor's max-norm."""
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "sparse")
    batch = _case_batch(cfg, B, case, seed=51)
    la, ga, gra, sa = _run_step(cfg, gs, batch, True, use_graph, steps=2)
    lb, gb, grb, sb_ = _run_step(cfg, gs, batch, False, use_graph, steps=2)
    assert la[0] == la[1]                                        # replay is idempotent without dropout
    assert abs(la[1] - lb[1]) < 2e-6 * max(1.0, abs(lb[1])), (la, lb)
    assert np.abs(ga - gb).max() < 2e-6, float(np.abs(ga - gb).max())
    for n in gra:
        assert _rel(gra[n], grb[n]) < 2e-5, (n, _rel(gra[n], grb[n]))
    p = sa.plan.cpu().numpy()
    assert p[0] == int(torch.clamp(batch["lengths"], 0, cfg["max_len"]).sum())
    F, T = cfg["d_inp"], cfg["max_len"]
    nz = (batch["src"][:, :, :F] != 0).any(-1).numpy()           # [T,B]
    lin = np.array([0 if not nz[:, b].any() else 1 + int(np.nonzero(nz[:, b])[0].max()) for b in range(B)])
    slack = int(np.maximum(lin - np.clip(batch["lengths"].numpy(), 0, T), 0).max())
    assert p[4] == slack, (p[:8], slack)                         # the input slack the forward kernel measured
    if case == "first_time_zero":
        assert slack == 1
    sa.close(); sb_.close()


def test_step_on_token_plan_with_dropout_is_deterministic():
    """Dropout on: the compact layout draws its masks by compact row, so values differ from the padded layout's; the sequence of
    losses must be finite, vary per replay and be reproducible."""
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "ones")
    batch = synth.make_batch(cfg, 64, seed=9)
    seqs = []
    for _ in range(2):
        losses, _, grads, st = _run_step(cfg, gs, batch, True, True, p_drop=0.2, steps=4)
        st.close()
        assert all(np.isfinite(losses)) and all(np.isfinite(g).all() for g in grads.values())
        seqs.append(losses)
    assert len(set(seqs[0])) == 4 and seqs[0] == seqs[1]


def test_dropout_gradient_consistency_on_token_plan():
    """With dropout on, forward and backward of one step must use the same masks on the compact layout: the directional derivative
    of the loss along the gradient (central difference on mlp_static / lin_value weights, same seed -> same masks) matches
    <g, d>."""
    from raindrop_amd.step import TrainStep
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "ones")
    batch = synth.make_batch(cfg, 32, seed=19)
    dv = {k: (None if v is None else v.to(DEV)) for k, v in batch.items()}
    m = build_ours(cfg, gs, DEV, 5).train()
    live = synth.live_parameter_names(cfg)
    named = dict(m.named_parameters())
    flat = dp.FlatGradAllReduce([(n, named[n]) for n in live])
    step = TrainStep(m, flat, dv, use_graph=False, token_plan=True, p_drop=0.2, autotune=False)   # eager: the seed cell is not bumped between runs
    assert step.plan is not None
    def run():
        step.seed_cell.zero_()                                    # every run bumps the cell: same masks for every evaluation
        return float(step.run())
    l0 = run(); torch.cuda.synchronize()
    for name in ("transformer_encoder.layers.0.linear1.weight", "ob_propagation_layer2.lin_value.weight"):
        w = named[name]
        g = w.grad.detach().clone()
        d = g / (g.norm() + 1e-30)
        eps = 1e-2
        with torch.no_grad():
            w.add_(eps * d); lp = run(); w.sub_(2 * eps * d); lm = run(); w.add_(eps * d)
        torch.cuda.synchronize()
        fd = (lp - lm) / (2 * eps)
        an = float((g * d).sum())
        assert abs(fd - an) < 0.05 * abs(an) + 1e-5, (name, fd, an, l0)
    step.close()
