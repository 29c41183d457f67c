"""GPU: the paper-faithful graph operator (use_beta branch, rd_graph_beta_fwd/_bwd) and the structure distance against
fixtures produced by the REFERENCE'S OWN classes (tests/golden/operators.npz `obpb_*`, tests/golden/beta_batched.npz):
pruned edge lists bit-exact (INT work), values to 1e-5, gradients to 2e-5 of their max-norm."""
import os

import numpy as np
import pytest
import torch

from oracle import restatement as O2
from raindrop_amd import ops, synth
from raindrop_amd.Ob_propagation import Observation_progation
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.asarray(a)).to(DEV)


def test_operator_use_beta_matches_reference_fixture():
    g = np.load(os.path.join(GOLDEN, "operators.npz"))
    op = Observation_progation(20, 20, n_nodes=6, ob_dim=4, heads=1)
    synth.fill_params_(op, seed=11)
    op = op.to(DEV)
    ei, ew = O2.build_graph(g["obp_adj"])
    y, (ei2, alpha) = op(_t(g["obp_x"]), p_t=_t(g["obp_p_t"]), edge_index=_t(ei), edge_weights=_t(ew), use_beta=True,
                         edge_attr=None, return_attention_weights=True)
    assert np.array_equal(ei2.cpu().numpy(), g["obpb_ei"])                      # pruned, re-ordered edge list: bit-exact
    assert np.abs(alpha.cpu().numpy() - g["obpb_alpha"]).max() < 1e-6
    assert np.abs(y.detach().cpu().numpy() - g["obpb_y"]).max() < 1e-5


def test_batched_beta_operator_forward_backward_and_distance():
    g = np.load(os.path.join(GOLDEN, "beta_batched.npz"))
    n, T, d, B = (int(v) for v in g["dims"])
    K = T * d
    op = Observation_progation(K, K, n_nodes=n, ob_dim=d, heads=1)
    synth.fill_params_(op, seed=21)
    op = op.to(DEV)
    ei, ew = O2.build_graph(g["adj"])
    X = _t(g["X"]).requires_grad_(True)
    # one batched launch over the B sample graphs (shared edges, per-sample features and time encodings)
    V = ops.linear(X.reshape(B * n, K), op.lin_value.weight, op.lin_value.bias, act=1).view(B, n, K)
    H = ops.linear(X.reshape(B * n, K), op.increase_dim.weight, op.increase_dim.bias, exact=True).view(B, n, T * 32)   # as the model does
    Y, ei2, alpha = ops.graph_beta(V, H, op.map_weights, _t(g["PT"]), _t(ei), _t(ew).reshape(1, -1), d)
    assert np.array_equal(ei2.cpu().numpy(), g["ei"])
    assert np.abs(alpha.cpu().numpy() - g["alpha"]).max() < 1e-6
    assert np.abs(Y.detach().cpu().numpy() - g["Y"]).max() < 1e-5
    grads = torch.autograd.grad((Y * _t(g["R"])).sum(), [X, op.lin_value.weight, op.lin_value.bias, op.increase_dim.weight,
                                                         op.increase_dim.bias, op.map_weights])
    for name, got in zip(["gX", "gWv", "gbv", "gWi", "gbi", "gmap"], grads):
        ref = g[name]
        assert np.abs(got.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-9, name
    # per-sample operator calls (the PyG-style API) give the same rows
    y0, (e0, a0) = op(X[0].detach(), p_t=_t(g["PT"][0]), edge_index=_t(ei), edge_weights=_t(ew), use_beta=True,
                      return_attention_weights=True)
    assert torch.equal(y0, Y[0].detach()) and torch.equal(e0, ei2[0])
    # structure distance of the returned scores (code/models_rd.py:345-346)
    dist = ops.structure_distance(alpha.t().contiguous())
    assert abs(float(dist) - float(g["distance"])) < 1e-6
    same = ops.structure_distance(alpha[:1].t().repeat(1, 7).contiguous())          # identical columns -> exactly 0
    assert float(same) == 0.0


def test_pruning_ties_keep_edge_order(monkeypatch):
    """All-ones adjacency (the shipped global_structure): every edge into one target has the same score, so half of the
    pruning decisions are ties.  The reference leaves them to `torch.argsort(descending=True)` WITHOUT stable=True, whose
    CPU tie order is an artefact of its introsort (e.g. 25 equal keys come back as [12, 24, 23, ...]) and differs between
    builds and devices: parity on ties is undefined upstream.  This implementation breaks ties by edge id (stable); the
    test pins that against the restatement run with a stable argsort."""
    n, T, d = 5, 4, 4
    K = T * d
    op = Observation_progation(K, K, n_nodes=n, ob_dim=d, heads=1)
    synth.fill_params_(op, seed=3)
    rng = np.random.default_rng(2)
    x = torch.from_numpy(rng.standard_normal((n, K)).astype(np.float32))
    p_t = torch.from_numpy(rng.standard_normal((T, 16)).astype(np.float32))
    ei, ew = O2.build_graph(np.ones((n, n), np.float32))
    real_argsort = torch.argsort
    monkeypatch.setattr(torch, "argsort", lambda t, *a, **k: real_argsort(t, *a, **dict(k, stable=True)))
    y_ref, (ei_ref, _) = O2.observation_propagation_beta(x, p_t, torch.from_numpy(ei), torch.from_numpy(ew),
                                                         op.lin_value.weight.detach(), op.lin_value.bias.detach(),
                                                         op.increase_dim.weight.detach(), op.increase_dim.bias.detach(),
                                                         op.map_weights.detach(), d)
    monkeypatch.undo()
    opd = op.to(DEV)
    y, (ei2, _) = opd(x.to(DEV), p_t=p_t.to(DEV), edge_index=_t(ei), edge_weights=_t(ew), use_beta=True,
                      return_attention_weights=True)
    assert np.array_equal(ei2.cpu().numpy(), ei_ref.numpy())
    assert np.abs(y.detach().cpu().numpy() - y_ref.numpy()).max() < 1e-5


@pytest.mark.parametrize("n,T", [(36, 215), (17, 600), (34, 60)])
def test_beta_operator_at_dataset_shapes(n, T, monkeypatch):
    """The use_beta operator at the reference's three dataset shapes (P12: 36 sensors x 215 steps, PAM: 17 x 600, P19: 34 x 60) on a
    sparse random structure (distinct edge scores: no pruning ties), forward and backward, against the restatement (O2, CPU
    autograd).  P12 and PAM need the per-direction LDS layout / the backward's time chunks (the five [N,T] arrays of the first
    version were 184 / 211 KB)."""
    d = 4
    K = T * d
    op = Observation_progation(K, K, n_nodes=n, ob_dim=d, heads=1)
    synth.fill_params_(op, seed=5)
    rng = np.random.default_rng(n * 1000 + T)
    adj = (rng.random((n, n)) < 0.3).astype(np.float32) * rng.uniform(0.5, 1.5, (n, n)).astype(np.float32)
    ei, ew = O2.build_graph(adj)
    x = torch.from_numpy((rng.standard_normal((n, K)) * 0.5).astype(np.float32))
    p_t = torch.from_numpy(rng.standard_normal((T, 16)).astype(np.float32))
    R = torch.from_numpy(rng.standard_normal((n, K)).astype(np.float32))
    real_argsort = torch.argsort
    monkeypatch.setattr(torch, "argsort", lambda t, *a, **k: real_argsort(t, *a, **dict(k, stable=True)))
    params = [op.lin_value.weight, op.lin_value.bias, op.increase_dim.weight, op.increase_dim.bias, op.map_weights]
    xr = x.clone().requires_grad_(True)
    pr = [p.detach().clone().requires_grad_(True) for p in params]
    y_ref, (ei_ref, a_ref) = O2.observation_propagation_beta(xr, p_t, torch.from_numpy(ei), torch.from_numpy(ew), *pr, d)
    g_ref = torch.autograd.grad((y_ref * R).sum(), [xr] + pr)
    monkeypatch.undo()
    opd = op.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    y, (ei2, alpha) = opd(xd, p_t=p_t.to(DEV), edge_index=_t(ei), edge_weights=_t(ew), use_beta=True, return_attention_weights=True)
    assert np.array_equal(ei2.cpu().numpy(), ei_ref.numpy())
    assert np.abs(alpha.cpu().numpy().ravel() - a_ref.detach().numpy().ravel()).max() < 1e-6
    assert np.abs(y.detach().cpu().numpy() - y_ref.detach().numpy()).max() < 2e-5
    g = torch.autograd.grad((y * R.to(DEV)).sum(), [xd, opd.lin_value.weight, opd.lin_value.bias, opd.increase_dim.weight,
                                                    opd.increase_dim.bias, opd.map_weights])
    for name, got, ref in zip(["x", "Wv", "bv", "Wi", "bi", "map"], g, g_ref):
        r = ref.numpy()
        assert np.abs(got.cpu().numpy() - r).max() <= 5e-5 * np.abs(r).max() + 1e-9, (name, float(np.abs(got.cpu().numpy() - r).max() / np.abs(r).max()))


def _beta_run(g, seed, large, monkeypatch):
    """The batched operator on fixture g (shared edge list), forward + every gradient; large: force the workspace form."""
    if large:
        monkeypatch.setenv("RD_BETA_LARGE", "1")
    else:
        monkeypatch.delenv("RD_BETA_LARGE", raising=False)
    n, T, d, B = (int(v) for v in g["dims"])
    K = T * d
    op = Observation_progation(K, K, n_nodes=n, ob_dim=d, heads=1)
    synth.fill_params_(op, seed=seed)
    op = op.to(DEV)
    ei, ew = O2.build_graph(g["adj"])
    X = _t(g["X"]).requires_grad_(True)
    V = ops.linear(X.reshape(B * n, K), op.lin_value.weight, op.lin_value.bias, act=1).view(B, n, K)
    H = ops.linear(X.reshape(B * n, K), op.increase_dim.weight, op.increase_dim.bias, exact=True).view(B, n, T * 32)   # as the model does
    ewd = _t(ew).reshape(1, -1).clone().requires_grad_(True)
    Y, ei2, alpha = ops.graph_beta(V, H, op.map_weights, _t(g["PT"]), _t(ei), ewd, d)
    grads = torch.autograd.grad((Y * _t(g["R"])).sum(), [X, op.lin_value.weight, op.lin_value.bias, op.increase_dim.weight,
                                                         op.increase_dim.bias, op.map_weights, ewd])
    monkeypatch.delenv("RD_BETA_LARGE", raising=False)
    return Y.detach(), ei2, alpha, [x.detach() for x in grads]


def test_workspace_form_equals_lds_form(monkeypatch):
    """rd_graph_beta_large.hip (state in a workspace, global-memory sort) against the LDS-staged kernels on a graph both take:
    same pruned edge list, and -- the sums run in the same order -- values and gradients to rounding (the two forms are different
    kernels, so the compiler may contract their multiply-adds differently)."""
    g = np.load(os.path.join(GOLDEN, "beta_batched.npz"))
    Ys, es, als, gs = _beta_run(g, 21, False, monkeypatch)
    Yl, el, all_, gl = _beta_run(g, 21, True, monkeypatch)
    assert torch.equal(es, el)
    assert float((als - all_).abs().max()) <= 1e-7
    assert float((Ys - Yl).abs().max()) <= 2e-6 * float(Ys.abs().max())
    for name, a, b in zip(["X", "Wv", "bv", "Wi", "bi", "map", "ew"], gs, gl):
        assert float((a - b).abs().max()) <= 5e-6 * float(a.abs().max()) + 1e-9, name
    # and against the reference's own outputs, like the LDS form
    assert np.array_equal(el.cpu().numpy(), g["ei"])
    assert np.abs(Yl.cpu().numpy() - g["Y"]).max() < 1e-5


def _assert_pruned_list_matches(ei_got, alpha_got, ei_ref, alpha_ref, max_swapped):
    """Pruned edge lists agree except where the reference's own scores are within rounding of each other: this implementation
    sums a score over the T steps, the reference averages the T*d repeated channels with torch's vectorised mean, so two scores
    a few ulps apart can come out in the other order (exact ties are excluded from the fixture by construction)."""
    assert np.abs(alpha_got - alpha_ref).max() <= 1e-6                                    # the sorted VALUES agree (scores are O(0.1))
    assert np.all(np.diff(alpha_got) <= 0)
    bad = np.nonzero((ei_got != ei_ref).any(axis=0))[0]
    assert bad.size <= max_swapped, bad.size
    score = {(int(s), int(t)): float(a) for s, t, a in zip(ei_ref[0], ei_ref[1], alpha_ref)}
    for q in bad:
        ref_score = score.get((int(ei_got[0, q]), int(ei_got[1, q])), float(alpha_ref[-1]))   # not kept upstream: a boundary swap
        assert abs(ref_score - float(alpha_ref[q])) <= 2e-6, (int(q), ref_score, float(alpha_ref[q]))


def test_large_graph_matches_reference_fixture(monkeypatch):
    """256 nodes, 13 360 edges (tests/golden/beta_large.npz, produced by the reference's Observation_progation): beyond the
    LDS-staged kernels (N <= 64, E <= 4096), so this runs the workspace form -- a 16 384-key sort in 4096-key chunks with
    global-memory merge steps, per-node lists over 6 680 kept edges -- forward and backward."""
    g = np.load(os.path.join(GOLDEN, "beta_large.npz"))
    n, T, d, B = (int(v) for v in g["dims"])
    assert ops._lib.load().rd_graph_beta_workspace_bytes(B, n, T * d, T, 13360) > 0
    # The score path (increase_dim -> beta -> mean over the steps -> sort key) runs on exact fp32 products in EVERY precision mode
    # (ops.linear(..., exact=True), round 5: index work is bit-exact by contract): the 13 k scores agree with the reference's to ~1e-8
    # and the pruned list comes out identical but for a handful of near-ties (the reference averages the repeated channels with torch's
    # vectorised mean, this implementation sums over the steps: a few ulps).  Round 4 let H follow the mode: ~1e-6 of rounding in
    # split-bf16 reordered ~1 % of the neighbours (1 311 of the fixture's 6 680 score gaps are below 1e-6) and the bound was 400.
    for mode, max_swapped in ((0, 8), (1, 8)):
        ops._lib.call("rd_set_precision", mode)
        try:
            Y, ei2, alpha, grads = _beta_run(g, 31, False, monkeypatch)
        finally:
            ops._lib.call("rd_set_precision", 1)
        for b in range(B):
            _assert_pruned_list_matches(ei2[b].cpu().numpy(), alpha[b].cpu().numpy(), g["ei"][b].astype(np.int64), g["alpha"][b].ravel(),
                                        max_swapped)
        assert np.abs(Y.cpu().numpy() - g["Y"]).max() < 2e-5 * max(1.0, np.abs(g["Y"]).max())
        for name, got in zip(["gX", "gWv", "gbv", "gWi", "gbi", "gmap"], grads[:6]):
            ref = g[name]
            assert np.abs(got.cpu().numpy() - ref).max() <= (5e-5 if mode == 0 else 2e-4) * np.abs(ref).max() + 1e-9, (mode, name)


def test_syn256_sized_graph_properties():
    """The SYN256 graph itself -- 256 sensors, all 65 536 edges, T = 512 steps, B = 2 -- through the workspace form: too large for
    a CPU oracle run inside a test, so the size-independent properties: the returned scores are sorted, the kept list holds
    exactly the int(E / 2) best edges by score (recomputed here from the saved per-step scores), every source's softmax weights sum
    to one at every step (out = sum of weights x V with V = 1 gives 1 wherever the source keeps an edge), gradients are finite and
    the weight gradient of an edge pruned in every sample is exactly zero."""
    rng = np.random.default_rng(9)
    n, T, d, B = 256, 512, 4, 2
    K = T * d
    E = n * n
    src, tgt = np.divmod(np.arange(E), n)
    ei = _t(np.stack([src, tgt]).astype(np.int64))
    ew = _t(rng.uniform(0.5, 1.5, (1, E)).astype(np.float32)).requires_grad_(True)
    V = torch.ones((B, n, K), device=DEV, requires_grad=True)
    H = _t((rng.standard_normal((B, n, T * 32)) * 0.5).astype(np.float32)).requires_grad_(True)
    mw = _t(rng.standard_normal((n, 16)).astype(np.float32))
    pt = _t(rng.standard_normal((B, T, 16)).astype(np.float32))
    Y, ei2, alpha = ops.graph_beta(V, H, mw, pt, ei, ew, d)
    Kk = E // 2
    assert tuple(ei2.shape) == (B, 2, Kk) and tuple(alpha.shape) == (B, Kk)
    assert bool((alpha[:, 1:] <= alpha[:, :-1]).all())
    # scores recomputed with torch (test side only): beta[b, n, t], score[e] = mean_t beta[tgt, t] * w[e]
    aa = torch.cat([mw[None, :, None, :].expand(B, n, T, 16), pt[:, None, :, :].expand(B, n, T, 16)], dim=-1)
    beta = (H.detach().view(B, n, T, 32) * aa).mean(-1)
    score = beta.mean(-1)[:, ei[1]] * ew.detach()                                   # [B, E]
    for b in range(B):
        kept = ei2[b, 0] * n + ei2[b, 1]
        assert kept.unique().numel() == Kk
        thr = float(alpha[b, -1])
        dropped = torch.ones(E, dtype=torch.bool, device=DEV); dropped[kept] = False
        tol = 1e-5 * float(score[b].abs().max())
        assert float(score[b][dropped].max()) <= thr + tol and float(score[b][kept].min()) >= thr - tol
        has_edge = torch.zeros(n, dtype=torch.bool, device=DEV); has_edge[ei2[b, 0]] = True
        want = has_edge[:, None].float().expand(n, K)
        assert float((Y[b].detach() - want).abs().max()) < 1e-4
    gV, gH, gw = torch.autograd.grad((Y * _t(rng.standard_normal((B, n, K)).astype(np.float32))).sum(), [V, H, ew])
    assert bool(torch.isfinite(gV).all()) and bool(torch.isfinite(gH).all()) and bool(torch.isfinite(gw).all())
    assert float(gV.abs().max()) > 0 and float(gH.abs().max()) > 0
    never = torch.ones(E, dtype=torch.bool, device=DEV)
    for b in range(B):
        never[ei2[b, 0] * n + ei2[b, 1]] = False
    assert float(gw[0][never].abs().max()) == 0.0                                  # pruned in every sample: no gradient at all


def test_graph_beta_rejects_malformed_input():
    n, T, d = 5, 4, 4
    K = T * d
    V = torch.zeros(2, n, K, device=DEV); H = torch.zeros(2, n, T * 32, device=DEV)
    mw = torch.zeros(n, 16, device=DEV); pt = torch.zeros(1, T, 16, device=DEV)
    ei = torch.tensor([[0, 1, 2], [1, 2, 7]], device=DEV)                      # endpoint 7 >= n
    ew = torch.ones(1, 3, device=DEV)
    with pytest.raises(IndexError):
        ops.graph_beta(V, H, mw, pt, ei, ew, d)
    ei_ok = torch.tensor([[0, 1, 2], [1, 2, 3]], device=DEV)
    with pytest.raises(ValueError):
        ops.graph_beta(V, H, torch.zeros(n, 8, device=DEV), pt, ei_ok, ew, d)  # map_weights not [N,16]
    with pytest.raises(ValueError):
        ops.graph_beta(V, H, mw, torch.zeros(3, T, 16, device=DEV), ei_ok, ew, d)   # p_t batch neither 1 nor B


@pytest.mark.parametrize("n,T,B", [(34, 60, 7), (36, 215, 3), (17, 600, 2), (6, 5, 4)])
def test_v2_kernels_equal_v1_bit_for_bit(n, T, B, monkeypatch):
    """Round 6's 16-wave kernels (every (edge, step) quantity formed once, lists by a wave per node, 16-byte accesses) keep the order
    of every sum of rounds 2-5's kernels (RD_BETA_V1=1): outputs, pruned lists, scores and ALL gradients -- d edge weight included --
    are the same bits, at the three dataset shapes and a tiny graph, on random sparse structures."""
    d = 4
    K = T * d
    rng = np.random.default_rng(n * 7 + T)
    adj = (rng.random((n, n)) < 0.4).astype(np.float32) * rng.uniform(0.5, 1.5, (n, n)).astype(np.float32)
    ei, ew = O2.build_graph(adj)
    op = Observation_progation(K, K, n_nodes=n, ob_dim=d, heads=1)
    synth.fill_params_(op, seed=9)
    op = op.to(DEV)
    X0 = torch.from_numpy((rng.standard_normal((B, n, K)) * 0.5).astype(np.float32)).to(DEV)
    PT = torch.from_numpy(rng.standard_normal((B, T, 16)).astype(np.float32)).to(DEV)
    R = torch.from_numpy(rng.standard_normal((B, n, K)).astype(np.float32)).to(DEV)
    outs = []
    for v1 in ("1", "0"):
        monkeypatch.setenv("RD_BETA_V1", v1)
        X = X0.clone().requires_grad_(True)
        V = ops.linear(X.reshape(B * n, K), op.lin_value.weight, op.lin_value.bias, act=1).view(B, n, K)
        H = ops.linear(X.reshape(B * n, K), op.increase_dim.weight, op.increase_dim.bias, exact=True).view(B, n, T * 32)
        ewd = _t(ew).reshape(1, -1).clone().requires_grad_(True)
        Y, ei2, alpha = ops.graph_beta(V, H, op.map_weights, PT, _t(ei), ewd, d)
        grads = torch.autograd.grad((Y * R).sum(), [X, op.lin_value.weight, op.increase_dim.weight, op.map_weights, ewd])
        outs.append((Y.detach(), ei2, alpha.detach(), [g_.detach() for g_ in grads]))
    monkeypatch.delenv("RD_BETA_V1", raising=False)
    (Y1, e1, a1, g1), (Y2, e2, a2, g2) = outs
    assert torch.equal(e1, e2) and torch.equal(a1, a2)
    assert torch.equal(Y1, Y2), float((Y1 - Y2).abs().max())
    for name, x, y in zip(["X", "Wv", "Wi", "map", "ew"], g1, g2):
        assert torch.equal(x, y), (name, float((x - y).abs().max()))
