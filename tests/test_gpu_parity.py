"""GPU: parity of the HIP path (through the C-ABI) against the oracle and the golden fixtures.
Tolerances: integer / boolean work bit-exact; fp32 logits within 1e-4 abs (north-star bound;
in practice ~1e-6); gradients within 1e-3 of their own max-norm."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import restatement as O2
from raindrop_amd import synth
from tests.helpers import BETA_CASES, MODEL_CASES, build_ours, case_inputs, golden_grad, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = {"x": 1.0}      # op-level tolerance multiplier, set per precision mode by the fixture below


@pytest.fixture(autouse=True, params=["bf16x3", "fp32"])
def precision_mode(request):
    """Every parity test runs in both arithmetic modes of the dense contractions: the default
    split-bf16 MFMA path (~2^-16 per product) and the exact fp32 MFMA path."""
    from raindrop_amd import _lib
    _lib.call("rd_set_precision", 1 if request.param == "bf16x3" else 0)
    TOL["x"] = 6.0 if request.param == "bf16x3" else 1.0
    yield request.param
    _lib.call("rd_set_precision", 1)


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _rel2(a, b):
    return float(np.linalg.norm((a - b).ravel().astype(np.float64)) / (np.linalg.norm(b.ravel().astype(np.float64)) + 1e-30))


def _grad_close(a, b, tol, name="", clear=None, tight=False):
    """Gradient comparison against a reference gradient (O1 fixture).  fp32 mode: max-norm relative error < tol.
    Split-bf16 mode: forward values differ from the reference at the 1e-5 level, so the few ReLU gates whose pre-activation lies
    within ~1e-5 of zero open / close differently.  ReLU's derivative is discontinuous there: ONE flipped gate moves the row of
    that unit in its Linear's weight gradient by |dh| |x| (measured: 1e-3..8e-3 of the matrix's L2 norm at a few hundred rows),
    and -- through the input gradient of that one row -- every gradient upstream by a little, while all other entries agree to
    ~1e-5.  Operator-level comparisons at a few hundred rows keep rounds 1-5's bounds for that mode (relative L2 < 2e-2,
    max-norm < 0.25: one flip among 360 tokens is 8e-3 / 7e-2; layout / indexing bugs produce O(1) errors).  `tight` -- the
    whole-model comparisons against the O1 fixtures at B >= 32 (numbers: tools/grad_gate_diag.py,
    profiles/r06_grad_gate_diag.txt):
      * all entries: relative L2 < 5e-3 and max-norm < 2e-2 (measured worst 3.3e-3 / 8.6e-3 over the fixtures);
      * `clear` (bool per sampled entry: the entry's unit is NOT on a ReLU fence by the fp32 restatement, tests/helpers
        .gate_unit_masks): relative L2 over those entries < 1e-3 for the encoder's and the head's Linear layers (measured 7e-6..
        3e-4; their own flipped rows were the whole 1e-3 error).  The two message-passing layers get no tighter bound from the
        mask: a flipped layer-2 gate perturbs one (sample, sensor) row of dZ1 and with it EVERY unit's row of dW1, and only
        ~10 of a sample's 34 sensor rows carry observations -- a few hundred effective rows in the sum (measured 3e-4..3.3e-3
        with or without the fenced units).
    The exact-fp32 mode keeps the tight max-norm bound on the very same kernels and code paths."""
    if TOL["x"] == 1.0:
        assert _rel(a, b) < tol, (name, _rel(a, b))
        return
    if not tight:
        assert _rel2(a, b) < 2e-2 and _rel(a, b) < 0.25, (name, _rel2(a, b), _rel(a, b))
        return
    assert _rel2(a, b) < 5e-3 and _rel(a, b) < 2e-2, (name, _rel2(a, b), _rel(a, b))
    if clear is not None and clear.any() and not name.startswith("ob_propagation"):
        assert _rel2(a[clear], b[clear]) < 1e-3, (name, "entries of units off the ReLU fences", _rel2(a[clear], b[clear]), int((~clear).sum()))


def _clear_entries(masks, name, param, stride, n):
    """bool [n]: sampled entry i (flat index i * stride of `param`) belongs to a unit that is not on a ReLU fence; None if the
    parameter feeds no ReLU."""
    if masks is None or name not in masks:
        return None
    cols = param.shape[1] if param.dim() == 2 else 1
    unit = (np.arange(n) * stride) // cols
    return ~masks[name][unit]


def _grad_close_masked(a, b, tol, name):
    return _grad_close(a, b, tol, name, tight=True)


@pytest.mark.parametrize("F,kind", [(1, "ones"), (5, "sparse"), (17, "ones"), (34, "sparse"), (36, "ones"),
                                    (64, "sparse"), (256, "sparse")])
def test_graph_build_bit_exact(F, kind):
    from raindrop_amd import ops
    rng = np.random.default_rng(F)
    a = np.ones((F, F), np.float32) if kind == "ones" else \
        (rng.random((F, F)) * (rng.random((F, F)) < 0.3)).astype(np.float32)
    adj, ei, ew = ops.graph_build(torch.from_numpy(a).to(DEV))
    ei_ref, ew_ref = O2.build_graph(a)
    assert np.array_equal(ei.cpu().numpy(), ei_ref)
    assert np.array_equal(ew.cpu().numpy(), ew_ref)
    patched = a.copy(); patched[np.arange(F), np.arange(F)] = 1
    assert np.array_equal(adj.cpu().numpy(), patched)
    gamma, ssum = ops.edge_softmax_dense(adj)
    g_ref, s_ref = O2.aggregate_coefficients(a)
    assert np.abs(gamma.cpu().numpy() - g_ref.numpy()).max() < 1e-6
    assert np.abs(ssum.cpu().numpy() - s_ref.numpy()).max() < 1e-6
    # list form agrees with the dense form
    ge, ss = ops.edge_softmax_list(ei, ew, F)
    assert np.abs(ss.cpu().numpy() - s_ref.numpy()).max() < 1e-6
    assert np.abs(ge.cpu().numpy() - g_ref.numpy()[ei_ref[0], ei_ref[1]]).max() < 1e-6


@pytest.mark.parametrize("cfg_name,B", [("TINY", 3), ("P19", 33), ("PAM", 2)])
def test_pe_and_mask(cfg_name, B):
    from raindrop_amd.models_rd import PositionalEncodingTF
    from raindrop_amd import _lib, ops
    import ctypes
    cfg = synth.make_config(cfg_name)
    b = synth.make_batch(cfg, B, seed=11)
    pe = PositionalEncodingTF(16, cfg["max_len"], 100)(b["times"].to(DEV))
    ref = O2.positional_encoding(b["times"], cfg["max_len"])
    assert np.abs(pe.cpu().numpy() - ref.numpy()).max() < 2e-6       # sin/cos within a few ulp
    shp = _lib.shape(B, cfg["max_len"], cfg["d_inp"], 4)
    z = torch.zeros(cfg["max_len"], B, cfg["d_inp"] * 4 + 16, device=DEV)
    mask = torch.zeros(B, cfg["max_len"], dtype=torch.bool, device=DEV)
    times_d, len_d, ts_d = b["times"].to(DEV), b["lengths"].to(DEV), ops.timescales(cfg["max_len"]).to(DEV)
    _lib.call("rd_pe_mask", ctypes.byref(shp), ops._ptr(times_d), ops._ptr(len_d), ops._ptr(ts_d),
              ops._ptr(z), ops._ptr(mask), ops._stream())
    assert np.array_equal(mask.cpu().numpy(), O2.padding_mask(b["lengths"].numpy(), cfg["max_len"]))
    assert float(z[:, :, : cfg["d_inp"] * 4].abs().max()) == 0.0     # message-passing columns untouched


@pytest.mark.parametrize("M,N,K,act", [(1, 1, 1, 0), (7, 5, 3, 1), (64, 64, 32, 0), (130, 186, 186, 1),
                                       (1000, 456, 152, 0), (257, 34, 6, 0), (513, 272, 152, 1),
                                       # long reductions (split-K weight gradients), ragged row counts
                                       (15360, 152, 272, 0), (5000, 272, 152, 0), (4099, 152, 152, 0),
                                       (3001, 456, 152, 0), (8704, 240, 240, 0), (1025, 64, 48, 0)])
def test_linear_fwd_bwd(M, N, K, act):
    from raindrop_amd import ops
    rng = np.random.default_rng(M * 7 + N)
    x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).requires_grad_(True)
    W = torch.from_numpy((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).requires_grad_(True)
    b = torch.from_numpy(rng.standard_normal((N,)).astype(np.float32)).requires_grad_(True)
    dy = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32))
    y_ref = torch.nn.functional.linear(x, W, b)
    y_ref = torch.relu(y_ref) if act else y_ref
    gx, gW, gb = torch.autograd.grad(y_ref, [x, W, b], dy)
    xd, Wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, W, b))
    y = ops.linear(xd, Wd, bd, act)
    hx, hW, hb = torch.autograd.grad(y, [xd, Wd, bd], dy.to(DEV))
    assert _rel(y.detach().cpu().numpy(), y_ref.detach().numpy()) < 1e-5 * TOL['x']
    assert _rel(hx.cpu().numpy(), gx.numpy()) < 1e-5 * TOL['x']
    assert _rel(hW.cpu().numpy(), gW.numpy()) < 2e-5 * TOL['x']
    assert _rel(hb.cpu().numpy(), gb.numpy()) < 2e-5


@pytest.mark.parametrize("cfg_name,B,kind,panel", [("TINY", 1, "sparse", None), ("TINY", 5, "ones", None), ("P19", 9, "sparse", None),
                                                   ("P12", 3, "ones", "0"), ("P12", 5, "sparse", "0"), ("PAM", 2, "sparse", "0"),
                                                   ("P12", 5, "sparse", "1"), ("P12", 9, "ones", "1"), ("PAM", 9, "sparse", "1"),
                                                   ("P12", 5, "sparse", "pc"), ("P12", 9, "ones", "pc"), ("PAM", 9, "sparse", "pc")])
def test_sensor_stage_vs_oracle(cfg_name, B, kind, panel, monkeypatch):
    """Observation embedding + both Observation_progation layers + PE concat, forward and backward,
    against the faithful (per-sample, per-edge) restatement of the reference.  P12 (K = 860: ragged in rows, columns and
    reduction chunks) and PAM (K = 2400, 34 rows: less than one row block) run the panel products of rd_gemm.hip with
    the d_ob = 4 scatter and the gate-mask epilogues -- `panel`: "0" the 64 x 128 workgroup tile, "1" round 6's 128 x 256 one, "pc" its
    128 x 128 producer / consumer one (ragged in both halves of their rows at 180 / 324 / 153 rows, in the last column block at 860
    and 2400 columns);
    P19 the fused kernels; TINY the tiled GEMM."""
    from raindrop_amd import _lib, ops
    if panel == "pc":                                     # the 128 x 128 producer / consumer form (waves 4-7 convert, waves 0-3 multiply)
        monkeypatch.setenv("RD_PANEL_PC", "1")
    elif panel is not None:
        monkeypatch.setenv("RD_PANEL_WIDE", panel)
        monkeypatch.setenv("RD_PANEL_PC", "0")
    cfg = synth.make_config(cfg_name)
    gs = synth.make_structure(cfg, kind)
    b = synth.make_batch(cfg, B, seed=21)
    T, F, d = cfg["max_len"], cfg["d_inp"], 4
    K = T * d
    names = ["R_u", "ob_propagation.lin_value.weight", "ob_propagation.lin_value.bias",
             "ob_propagation_layer2.lin_value.weight", "ob_propagation_layer2.lin_value.bias"]
    shapes = [(1, F * d), (K, K), (K,), (K, K), (K,)]
    p = {n: synth.param_values(n, s, seed=5).requires_grad_(True) for n, s in zip(names, shapes)}
    # oracle: faithful per-edge evaluation of just this stage
    h = torch.relu(torch.repeat_interleave(b["src"][:, :, :F], d, dim=-1) * p["R_u"])
    ei_np, ew_np = O2.build_graph(gs.numpy())
    ei, ew = torch.from_numpy(ei_np), torch.from_numpy(ew_np)
    cols = []
    for u in range(B):
        x = h[:, u, :].reshape(T, F, d).permute(1, 0, 2).reshape(F, K)
        x, a1 = O2.observation_propagation(x, ei, ew, p[names[1]], p[names[2]])
        x, _ = O2.observation_propagation(x, ei, a1.squeeze(-1), p[names[3]], p[names[4]])
        cols.append(x.view(F, T, d).permute(1, 0, 2).reshape(T, F * d))
    out_ref = torch.stack(cols, dim=1)
    dz = torch.from_numpy(np.random.default_rng(3).standard_normal((T, B, F * d + 16)).astype(np.float32))
    g_ref = torch.autograd.grad(out_ref, [p[n] for n in names], dz[:, :, : F * d])
    # device
    pd = {n: t.detach().to(DEV).requires_grad_(True) for n, t in p.items()}
    adj, _, _ = ops.graph_build(gs.to(DEV))
    _, ssum = ops.edge_softmax_dense(adj)
    shp = _lib.shape(B, T, F, d)
    z, mask = ops.sensor_stage(b["src"].to(DEV), b["times"].to(DEV), b["lengths"].to(DEV),
                               ops.timescales(T).to(DEV), ssum, pd[names[0]], pd[names[1]], pd[names[2]],
                               pd[names[3]], pd[names[4]], shp)
    g = torch.autograd.grad(z, [pd[n] for n in names], dz.to(DEV))
    assert np.abs(z[:, :, : F * d].detach().cpu().numpy() - out_ref.detach().numpy()).max() < 2e-5 * TOL['x']
    pe_ref = O2.positional_encoding(b["times"], T)
    assert np.abs(z[:, :, F * d:].detach().cpu().numpy() - pe_ref.numpy()).max() < 2e-6
    assert np.array_equal(mask.cpu().numpy(), O2.padding_mask(b["lengths"].numpy(), T))
    for n, a, r in zip(names, g, g_ref):
        _grad_close(a.cpu().numpy(), r.numpy(), 1e-4, n)


@pytest.mark.parametrize("graph_step", [False, True], ids=["operators", "module_graph"])
@pytest.mark.parametrize("name", MODEL_CASES)
def test_model_vs_golden(name, graph_step):
    """Whole Raindrop_v2 forward + CE + backward against the fixtures produced by the reference -- operator by operator under
    autograd, and the way an unmodified script gets it by default (the captured forward / backward behind model.forward,
    raindrop_amd/graph_module.py; shapes outside its envelope fall back to the operators by themselves)."""
    g, meta = load_golden(name)
    cfg, gs, batch = case_inputs(meta)
    m = build_ours(cfg, gs, DEV, meta["param_seed"], float(meta.get("param_scale", 1.0)))
    m.graph_step = graph_step
    m.train()
    dv = {k: (None if v is None else v.to(DEV)) for k, v in batch.items()}
    logits, distance, third = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
    assert third is None and float(distance) == float(g["distance"]) == 0.0
    loss = torch.nn.functional.cross_entropy(logits, dv["y"])
    loss.backward()
    assert np.abs(logits.detach().cpu().numpy() - g["logits"]).max() < 1e-4
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    params = dict(m.named_parameters())
    live = [str(x) for x in g["live"]]
    assert sorted(n for n, p in params.items() if p.grad is not None) == sorted(live)
    from tests.helpers import gate_unit_masks
    tight = meta["cfg"] in ("P19", "P12") and meta["batch"] >= 32          # TINY / PAM / SYN256 fixtures are 2-4 samples: a flip weighs more
    masks = gate_unit_masks(meta) if (TOL["x"] != 1.0 and tight) else None
    for n in live:
        exp, got = golden_grad(g, n, params[n].grad)
        _grad_close(got, exp, 1e-3, n, _clear_entries(masks, n, params[n], int(g["gradstride/" + n]), exp.size), tight=tight)
        gn = float(g["gradnorm/" + n])
        assert abs(params[n].grad.double().norm().item() - gn) <= 1e-3 * gn + 1e-12, n
    m.eval()
    with torch.no_grad():
        le, _, _ = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
    assert np.abs(le.cpu().numpy() - g["logits_eval"]).max() < 1e-4
    # graph artefacts the model cached are the reference's, bit for bit
    gr = m._graph(torch.device(DEV))
    assert np.array_equal(gr["edge_index"].cpu().numpy(), g["edge_index"])
    assert np.array_equal(gr["edge_weights"].cpu().numpy(), g["edge_weights"])


@pytest.mark.parametrize("name", BETA_CASES)
def test_model_use_beta_vs_golden(name, precision_mode):
    """Raindrop_v2(use_beta=True, compute_distance=True): the paper's branch of the model (layer 1 = use_beta operator with the
    sample's positional encoding as p_t, per-sample pruning, layer 2 on the surviving edges) against fixtures produced by the
    reference's own forward with its `use_beta = False` literal (code/models_rd.py:317) flipped.  Logits 1e-4, loss 1e-5, the
    structure distance (non-zero on this branch) 1e-5 relative, every gradient -- incl. map_weights / increase_dim, which only
    this branch trains."""
    g, meta = load_golden(name)
    cfg, gs, batch = case_inputs(meta)
    m = build_ours(cfg, gs, DEV, meta["param_seed"], use_beta=True, compute_distance=True)
    m.train()
    dv = {k: (None if v is None else v.to(DEV)) for k, v in batch.items()}
    logits, distance, third = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
    assert third is None
    loss = torch.nn.functional.cross_entropy(logits, dv["y"])
    loss.backward()
    assert np.abs(logits.detach().cpu().numpy() - g["logits"]).max() < 1e-4
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    assert abs(float(distance) - float(g["distance"])) <= 1e-5 * float(g["distance"]) + 1e-7, (float(distance), float(g["distance"]))
    params = dict(m.named_parameters())
    live = [str(x) for x in g["live"]]
    assert sorted(n for n, p in params.items() if p.grad is not None) == sorted(live)
    for n in live:
        exp, got = golden_grad(g, n, params[n].grad)
        _grad_close(got, exp, 1e-3, n)
        gn = float(g["gradnorm/" + n])
        assert abs(params[n].grad.double().norm().item() - gn) <= 1e-3 * gn + 1e-12, n
    # default keywords reproduce the reference's literals: use_beta off, distance the exact constant 0
    m2 = build_ours(cfg, gs, DEV, meta["param_seed"])
    assert not m2.use_beta and not m2.compute_distance
    with torch.no_grad():
        _, d0, _ = m2(dv["src"], dv["static"], dv["times"], dv["lengths"])
    assert float(d0) == 0.0


def test_legacy_raindrop_vs_golden(precision_mode):
    """The legacy `Raindrop` model (code/models_rd.py:46-191) on the HIP operators -- input encoder + scale + dropout, batched
    TransformerConv over the [T, 36] step matrices, 36-wide PE, temporal encoder (D = 108, four heads of 27), masked mean, head --
    against the reference's own forward / backward at the shape its forward hard-codes (215 steps, 36 sensors)."""
    import json, os
    from raindrop_amd.models_rd import Raindrop
    from tests.helpers import GOLDEN, zero_dropout
    g = np.load(os.path.join(GOLDEN, "legacy_v1.npz"), allow_pickle=False)
    meta = json.loads(str(g["meta"]))
    gs = synth.make_structure(dict(d_inp=meta["d_inp"]), "sparse", seed=meta["structure_seed"])
    m = Raindrop(meta["d_inp"], meta["d_model"], meta["nhead"], meta["nhid"], meta["nlayers"], 0.3, meta["max_len"], meta["d_static"],
                 100, 0.5, "mean", 2, gs)
    synth.fill_params_(m, seed=meta["param_seed"])
    zero_dropout(m)
    m = m.to(DEV).train()
    cfg = dict(max_len=meta["max_len"], d_inp=meta["d_inp"], static=True, d_static=meta["d_static"], n_classes=2)
    b = synth.make_batch(cfg, meta["batch"], seed=meta["batch_seed"], density=meta["density"])
    dv = {k: (None if v is None else v.to(DEV)) for k, v in b.items()}
    logits, distance, third = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
    assert third is None and float(distance) == float(g["distance"]) == 0.0
    loss = torch.nn.functional.cross_entropy(logits, dv["y"])
    loss.backward()
    assert np.abs(logits.detach().cpu().numpy() - g["logits"]).max() < 1e-4
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    params = dict(m.named_parameters())
    live = [str(x) for x in g["live"]]
    assert sorted(n for n, p in params.items() if p.grad is not None) == sorted(live)
    for n in live:
        exp, got = golden_grad(g, n, params[n].grad)
        _grad_close(got, exp, 1e-3, n)
    m.eval()
    with torch.no_grad():
        le, _, _ = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
    assert np.abs(le.cpu().numpy() - g["logits_eval"]).max() < 1e-4
    # input dropout (code/models_rd.py:134) on the device RNG: deterministic per call counter, and it changes the output
    m.train()
    m.dropout.p = 0.3
    torch.manual_seed(5); m._drop_calls = 0
    a1 = m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0]
    torch.manual_seed(5); m._drop_calls = 0
    a2 = m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0]
    assert torch.equal(a1, a2) and not torch.equal(a1, logits)


def test_batched_edge_ops_match_single_graph_calls():
    """rd_edge_softmax_list_batched == B calls of rd_edge_softmax_list (bit for bit); rd_edge_gamma_dense == the scatter of the
    per-edge coefficients (duplicate edges added in edge order); rd_aggregate_batched == B calls of rd_aggregate, fwd + bwd."""
    from raindrop_amd import ops
    rng = np.random.default_rng(5)
    B, N, E, C = 5, 23, 140, 37
    ei = torch.from_numpy(rng.integers(0, N, size=(B, 2, E))).to(DEV)
    ei[:, :, 7] = ei[:, :, 3]                                        # a duplicate edge per graph
    w = torch.from_numpy(rng.standard_normal((B, E)).astype(np.float32)).to(DEV)
    gb, sb = ops.edge_softmax_list_batched(ei, w, N, norm_row=1)
    for b in range(B):
        g1, s1 = ops.edge_softmax_list(ei[b], w[b], N, norm_row=1)
        assert torch.equal(gb[b], g1) and torch.equal(sb[b], s1)
    gs_, ss_ = ops.edge_softmax_list_batched(ei[0], w, N, norm_row=0)  # one shared list, per-graph weights, source-normalised
    for b in range(B):
        g1, s1 = ops.edge_softmax_list(ei[0], w[b], N, norm_row=0)
        assert torch.equal(gs_[b], g1) and torch.equal(ss_[b], s1)
    dense = ops.edge_gamma_dense(ei[0], gb[0], N).cpu().numpy()
    ref = np.zeros((N, N), np.float32)
    e0, g0 = ei[0].cpu().numpy(), gb[0].cpu().numpy()
    for e in range(E):
        ref[e0[0, e], e0[1, e]] = np.float32(ref[e0[0, e], e0[1, e]] + g0[e])
    assert np.array_equal(dense, ref)
    gam = torch.from_numpy(dense).to(DEV)
    V = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(DEV).requires_grad_(True)
    S = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(DEV).requires_grad_(True)
    R = torch.from_numpy(rng.standard_normal((B, N, C)).astype(np.float32)).to(DEV)
    out = ops.aggregate_batched(gam, V, S)
    gV, gS = torch.autograd.grad((out * R).sum(), [V, S])
    for b in range(B):
        v1 = V[b].detach().clone().requires_grad_(True)
        s1 = S[b].detach().clone().requires_grad_(True)
        o1 = ops.aggregate(gam, v1, s1)
        g1 = torch.autograd.grad((o1 * R[b]).sum(), [v1, s1])
        assert torch.allclose(out[b], o1, rtol=0, atol=1e-6) and torch.allclose(gV[b], g1[0], rtol=0, atol=1e-6)
        assert torch.equal(gS[b], g1[1])
    ref_out = np.einsum("ji,bjc->bic", dense.astype(np.float64), V.detach().cpu().numpy().astype(np.float64)) + S.detach().cpu().numpy()
    assert np.abs(out.detach().cpu().numpy() - ref_out).max() < 1e-4


def test_operator_goldens_on_device():
    import os
    from raindrop_amd.Ob_propagation import Observation_progation
    from raindrop_amd.transformer_conv import TransformerConv
    from tests.helpers import GOLDEN
    g = np.load(os.path.join(GOLDEN, "operators.npz"))
    t = lambda a: torch.from_numpy(a).to(DEV)
    op = Observation_progation(20, 20, n_nodes=6, ob_dim=4, heads=1)
    synth.fill_params_(op, seed=11)
    op = op.to(DEV)
    ei, ew = O2.build_graph(g["obp_adj"])
    y, (ei_o, alpha) = op(t(g["obp_x"]), p_t=t(g["obp_p_t"]), edge_index=t(ei), edge_weights=t(ew),
                          use_beta=False, edge_attr=None, return_attention_weights=True)
    assert np.abs(y.detach().cpu().numpy() - g["obp_y"]).max() < 1e-5 * TOL['x']
    assert np.array_equal(alpha.cpu().numpy(), g["obp_alpha"]) and np.array_equal(ei_o.cpu().numpy(), g["obp_ei"])
    tc = TransformerConv(9, 12, heads=1)
    synth.fill_params_(tc, seed=12)
    tc = tc.to(DEV)
    ei2, ew2 = O2.build_graph(g["tc_adj"])
    y2, (_, a2) = tc(t(g["tc_x"]), edge_index=t(ei2), edge_weights=t(ew2), edge_attr=None,
                     return_attention_weights=True)
    assert np.abs(y2.detach().cpu().numpy() - g["tc_y"]).max() < 1e-5 * TOL['x']
    assert np.abs(a2.cpu().numpy() - g["tc_alpha"]).max() < 1e-6


def _enc_params(D, nhid, seed):
    from raindrop_amd import ops
    shapes = {"self_attn.in_proj_weight": (3 * D, D), "self_attn.in_proj_bias": (3 * D,),
              "self_attn.out_proj.weight": (D, D), "self_attn.out_proj.bias": (D,),
              "linear1.weight": (nhid, D), "linear1.bias": (nhid,), "linear2.weight": (D, nhid),
              "linear2.bias": (D,), "norm1.weight": (D,), "norm1.bias": (D,), "norm2.weight": (D,), "norm2.bias": (D,)}
    return {n: synth.param_values("L." + n, shapes[n], seed=seed) for n in ops.ENC_PARAM_NAMES}


@pytest.mark.parametrize("T,B,F,nhead", [(7, 3, 5, 2), (60, 6, 34, 2), (215, 2, 36, 2), (130, 2, 17, 2), (64, 3, 12, 4),
                                         (70, 2, 256, 2), (40, 3, 60, 2)])
def test_encoder_layer_vs_oracle(T, B, F, nhead):
    """One TransformerEncoderLayer (attention with key-padding mask, LN, FFN) forward and backward
    against the torch restatement; T=215/130 exercise the multi-tile online softmax, F=256 / 60 (head_dim 520 / 128) the
    materialised-score attention of wide heads."""
    from raindrop_amd import _lib, ops
    D, nhid = F * 4 + 16, 2 * F * 4
    rng = np.random.default_rng(T * 31 + B)
    x = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32))
    lengths = torch.from_numpy(rng.integers(1, T + 1, size=B)).long()
    lengths[0] = T
    mask = torch.from_numpy(O2.padding_mask(lengths.numpy(), T))
    dy = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32))
    p = _enc_params(D, nhid, seed=T)
    xr = x.clone().requires_grad_(True)
    pr = {("L." + n): t.clone().requires_grad_(True) for n, t in p.items()}
    y_ref = O2.encoder_layer(xr, mask, pr, "L.", nhead)
    g_ref = torch.autograd.grad(y_ref, [xr] + [pr["L." + n] for n in ops.ENC_PARAM_NAMES], dy)
    xd = x.to(DEV).requires_grad_(True)
    pd = [p[n].to(DEV).requires_grad_(True) for n in ops.ENC_PARAM_NAMES]
    shp = _lib.shape(B, T, F, 4, nhead=nhead, nhid=nhid)
    y = ops.encoder_layer(xd, mask.to(DEV), shp, 0, 0.0, 0, pd)
    g = torch.autograd.grad(y, [xd] + pd, dy.to(DEV))
    assert np.abs(y.detach().cpu().numpy() - y_ref.detach().numpy()).max() < 5e-5 * TOL['x']
    for name, a, r in zip(["x"] + list(ops.ENC_PARAM_NAMES), g, g_ref):
        _grad_close(a.cpu().numpy(), r.numpy(), 2e-4, name)


@pytest.mark.parametrize("T,B,F,nhead", [(60, 9, 34, 2), (33, 4, 34, 2), (64, 3, 12, 4), (215, 2, 36, 2), (16, 5, 17, 2),
                                         (215, 5, 36, 2), (65, 3, 17, 2), (130, 4, 12, 4), (300, 3, 20, 1)])
def test_attention_core_vs_float64(T, B, F, nhead):
    """The attention core by itself (rd_attention_fwd / rd_attention_bwd: the kernels the encoder layer runs between in_proj and
    out_proj) against float64 torch: softmax(q k^T / sqrt(hd) with padded keys at -inf) v per head, and its gradient.  No ReLU
    gate lives in this sub-graph, so -- unlike the whole-model gradient checks, which a flipped gate can move by a percent -- the
    split-bf16 kernels (single tile T <= 64, multi tile above: 215 = P12, 65 = one key past a tile, head_dim 42 / 16 / 96, a
    20-step sample whose later key tiles are skipped as dead) are held to 2e-4 of each tensor's max-norm directly against the
    oracle arithmetic; the exact-fp32 mode to 2e-5."""
    from raindrop_amd import _lib
    D = F * 4 + 16
    hd = D // nhead
    rng = np.random.default_rng(T * 13 + B)
    qkv = torch.from_numpy(rng.standard_normal((T, B, 3 * D)).astype(np.float32))
    lengths = torch.from_numpy(rng.integers(1, T + 1, size=B)).long()
    lengths[0] = T
    if B > 2:
        lengths[1] = min(T, 20)
    mask = torch.from_numpy(O2.padding_mask(lengths.numpy(), T))
    dout = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32))
    # ---- float64 reference (torch semantics: F.multi_head_attention_forward, key_padding_mask -> -inf) ----
    q64 = qkv.double().requires_grad_(True)
    q, k, v = q64[..., :D], q64[..., D:2 * D], q64[..., 2 * D:]
    sh = lambda t: t.reshape(T, B, nhead, hd).permute(1, 2, 0, 3)                       # [B,H,T,hd]
    S = (sh(q) / np.sqrt(hd)) @ sh(k).transpose(-1, -2)
    S = S.masked_fill(mask[:, None, None, :], float("-inf"))
    out64 = (torch.softmax(S, -1) @ sh(v)).permute(2, 0, 1, 3).reshape(T, B, D)
    (g64,) = torch.autograd.grad(out64, q64, dout.double())
    # ---- device ----
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    qd, md, dd = qkv.to(DEV), mask.to(DEV), dout.to(DEV)
    out = torch.zeros(T, B, D, device=DEV); lse = torch.zeros(B, nhead, T, device=DEV)
    dqkv = torch.zeros(T, B, 3 * D, device=DEV); ws = torch.zeros(B, nhead, T, device=DEV)
    shp = _lib.shape(B, T, F, 4, nhead=nhead, nhid=2 * F * 4)
    _lib.call("rd_attention_fwd", ctypes.byref(shp), 0, P(qd), P(md), 0.0, 0, P(out), P(lse), None)
    _lib.call("rd_attention_bwd", ctypes.byref(shp), 0, P(qd), P(md), 0.0, 0, P(out), P(lse), P(dd), P(dqkv), P(ws), None)
    torch.cuda.synchronize()
    tol = 2e-4 if TOL["x"] != 1.0 else 2e-5
    # padded QUERY rows are computed too (torch does): compare everything
    assert _rel(out.cpu().numpy(), out64.detach().numpy()) < tol, _rel(out.cpu().numpy(), out64.detach().numpy())
    assert _rel(dqkv.cpu().numpy(), g64.numpy()) < tol, _rel(dqkv.cpu().numpy(), g64.numpy())


@pytest.mark.parametrize("T,B,F,nhead,p_drop", [(215, 4, 36, 2, 0.2), (130, 3, 17, 2, 0.2), (70, 2, 12, 4, 0.0)])
def test_multi_tile_attention_split_bf16_matches_fp32_kernels(T, B, F, nhead, p_drop, precision_mode, monkeypatch):
    """T > 64: the split-bf16 flash kernels (k_attn_fwd_b16 / k_attn_bwd_dq_b16 / k_attn_bwd_dkv_b16, dead key tiles skipped)
    against the exact-fp32 flash kernels they replace in the bf16 modes (RD_ATTN_B16_MT=0), WITH attention dropout on: the
    keep mask is a function of (seed, site, head, query, key) only, so both forms drop the same probabilities and the outputs,
    the saved log-sum-exp and the gradients agree to the product precision.  An arbitrary (non-prefix) key mask is used: the
    C-ABI takes any [B,T] mask, not only padding."""
    if precision_mode == "fp32":
        pytest.skip("exact-fp32 mode runs the fp32 kernels only")
    from raindrop_amd import _lib
    D = F * 4 + 16
    rng = np.random.default_rng(T * 5 + B)
    qkv = torch.from_numpy(rng.standard_normal((T, B, 3 * D)).astype(np.float32)).to(DEV)
    dout = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    m = rng.random((B, T)) < 0.3
    m[0, :] = False
    m[1, 64:] = True                     # sample 1: every key tile after the first is dead
    m[1, 5] = False
    m[:, 0] = False
    mask = torch.from_numpy(m).to(DEV)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    shp = _lib.shape(B, T, F, 4, nhead=nhead, nhid=2 * F * 4)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RD_ATTN_B16_MT", mode)
        out = torch.zeros(T, B, D, device=DEV); lse = torch.zeros(B, nhead, T, device=DEV)
        dqkv = torch.zeros(T, B, 3 * D, device=DEV); ws = torch.zeros(B, nhead, T, device=DEV)
        _lib.call("rd_attention_fwd", ctypes.byref(shp), 0, P(qkv), P(mask), p_drop, 77, P(out), P(lse), None)
        _lib.call("rd_attention_bwd", ctypes.byref(shp), 0, P(qkv), P(mask), p_drop, 77, P(out), P(lse), P(dout), P(dqkv), P(ws), None)
        torch.cuda.synchronize()
        res[mode] = (out.cpu().numpy(), lse.cpu().numpy(), dqkv.cpu().numpy())
    monkeypatch.delenv("RD_ATTN_B16_MT")
    tol = 2e-4 * (4.0 if precision_mode == "bf16" else 1.0)
    for name, a, b in zip(("out", "lse", "dqkv"), res["1"], res["0"]):
        assert np.isfinite(a).all(), name
        assert _rel(a, b) < tol, (name, _rel(a, b))


@pytest.mark.parametrize("T,B,p_drop,F", [(60, 6, 0.0, 34), (60, 150, 0.2, 34), (60, 256, 0.2, 34), (33, 7, 0.2, 34), (60, 137, 0.0, 34),
                                          (215, 8, 0.2, 36), (40, 45, 0.2, 36), (50, 9, 0.2, 35)])
def test_fused_row_local_chains_match_row_block_products(T, B, p_drop, F, precision_mode, monkeypatch):
    """rd_encfuse.hip (out_proj + LayerNorm1 + FFN + LayerNorm2 in one launch, and its backward chain) against the three
    row-block launches per direction it replaces, IN THE SAME ARITHMETIC with the same dropout quads (dropout ON): the two paths
    must agree to summation order (LayerNorm row sums; per-block LayerNorm partials: 32- vs 48-row blocks) in the output and
    the gradients.  B = 150 / 137 (9000 / 8220 rows) make the kernels pick their 48-row blocks on a 256-CU device, B = 256
    (15360 rows) two rounds of 32-row blocks.  F = 34 / 36 are the widths compiled in (P19: 152 x 272, P12: 160 x 288), F = 35
    runs the runtime-width instantiation."""
    if precision_mode == "fp32":
        pytest.skip("the fused chains are bf16-mode kernels")
    from raindrop_amd import _lib, ops
    nhead = 2
    D, nhid = F * 4 + 16, 2 * F * 4
    rng = np.random.default_rng(T * 7 + B)
    x = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    lengths = torch.from_numpy(rng.integers(1, T + 1, size=B)).long()
    lengths[0] = T
    mask = torch.from_numpy(O2.padding_mask(lengths.numpy(), T)).to(DEV)
    dy = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    p = _enc_params(D, nhid, seed=T)
    shp = _lib.shape(B, T, F, 4, nhead=nhead, nhid=nhid)
    res = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("RD_ENC_FUSE", fuse)
        xd = x.clone().requires_grad_(True)
        pd = [p[n].to(DEV).requires_grad_(True) for n in ops.ENC_PARAM_NAMES]
        y = ops.encoder_layer(xd, mask, shp, 1, p_drop, 77, pd)
        g = torch.autograd.grad(y, [xd] + pd, dy)
        res[fuse] = (y.detach().cpu().numpy(), [t.cpu().numpy() for t in g])
    # same products, same dropout quads; the LayerNorm row sums are grouped differently (16-lane butterflies over three quads per
    # lane in the fused chains, one 64-lane tree in the row-block kernels): rounding-level differences only
    assert _rel(res["1"][0], res["0"][0]) < 2e-6, _rel(res["1"][0], res["0"][0])
    for name, a, b in zip(["x"] + list(ops.ENC_PARAM_NAMES), res["1"][1], res["0"][1]):
        assert _rel(a, b) < 2e-5, (name, _rel(a, b))


@pytest.mark.parametrize("T,B", [(60, 37), (60, 256), (33, 5)])
def test_encoder_tile_weight_gradients_match_split_k(T, B, precision_mode, monkeypatch):
    """The streamed weight gradients (rd_tile_wgrad.hip: operands exported as split-bf16 row tiles by the row-block
    products, one launch for the layer's four dW/db) against the tiled split-K products they replace, in the SAME
    split-bf16 arithmetic and with the same dropout masks: only the summation order differs, so every entry must agree
    within 2e-5 of the tensor's max-norm (the LayerNorm gradients too: their partial sums are grouped by 64 rows on the
    tile path, where the LayerNorm backward is a prologue of the input-gradient products, and by 16 on the other).
    T*B = 2220 / 165 rows end in a partial 32-row chunk."""
    if precision_mode != "bf16x3":
        pytest.skip("the tile stream exists in split-bf16 mode only")
    from raindrop_amd import _lib, ops
    F, nhead = 34, 2
    D, nhid = F * 4 + 16, 2 * F * 4
    rng = np.random.default_rng(T + B)
    x = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    lengths = torch.from_numpy(rng.integers(1, T + 1, size=B)).long()
    mask = torch.from_numpy(O2.padding_mask(lengths.numpy(), T)).to(DEV)
    dy = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    p = _enc_params(D, nhid, seed=B)
    shp = _lib.shape(B, T, F, 4, nhead=nhead, nhid=nhid)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RD_TILE_WGRAD", mode)
        xd = x.clone().requires_grad_(True)
        pd = [p[n].to(DEV).requires_grad_(True) for n in ops.ENC_PARAM_NAMES]
        y = ops.encoder_layer(xd, mask, shp, 1, 0.2, 1234, pd)
        g = torch.autograd.grad(y, [xd] + pd, dy)
        torch.cuda.synchronize()
        out[mode] = (y.detach().cpu().numpy(), [t.cpu().numpy() for t in g])
    monkeypatch.delenv("RD_TILE_WGRAD")
    assert np.array_equal(out["1"][0], out["0"][0])
    for name, a, r in zip(["x"] + list(ops.ENC_PARAM_NAMES), out["1"][1], out["0"][1]):
        if name == "x":
            assert np.abs(a - r).max() <= 1e-5 * np.abs(r).max(), name      # LayerNorm backward inside the fused chain: same arithmetic,
        else:                                                            # row sums grouped differently (16-lane butterflies vs one 64-lane tree)
            assert np.abs(a - r).max() <= 2e-5 * np.abs(r).max(), (name, float(np.abs(a - r).max() / np.abs(r).max()))


@pytest.mark.parametrize("T,B,F", [(16, 65, 68), (40, 27, 68)])
def test_wide_encoder_tile_weight_gradients_match_split_k(T, B, F, precision_mode, monkeypatch):
    """Widths beyond the row-block kernels (D = 288, nhid = 544: the panel / tiled GEMM path SYN256 takes): the layer's four weight
    gradients through the stand-alone conversion pass + tile stream (rd_tiles_export.hip, round 4) against the four split-K GEMMs
    they replace (RD_TILE_WGRAD_GENERIC=0), same split-bf16 arithmetic, same dropout masks: identical output and input gradient,
    weight gradients within 2e-5 of the tensor's max-norm.  T*B = 1040 / 1080 rows: a partial last 32-row chunk, and
    column counts (288, 544, 864) that are not multiples of the 64-column conversion blocks."""
    if precision_mode != "bf16x3":
        pytest.skip("the tile stream exists in the bf16 modes only")
    from raindrop_amd import _lib, ops
    nhead = 4
    D, nhid = F * 4 + 16, 2 * F * 4
    rng = np.random.default_rng(T * B)
    x = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    lengths = torch.from_numpy(rng.integers(1, T + 1, size=B)).long()
    mask = torch.from_numpy(O2.padding_mask(lengths.numpy(), T)).to(DEV)
    dy = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    p = _enc_params(D, nhid, seed=B)
    shp = _lib.shape(B, T, F, 4, nhead=nhead, nhid=nhid)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RD_TILE_WGRAD_GENERIC", mode)
        xd = x.clone().requires_grad_(True)
        pd = [p[n].to(DEV).requires_grad_(True) for n in ops.ENC_PARAM_NAMES]
        y = ops.encoder_layer(xd, mask, shp, 1, 0.2, 1234, pd)
        g = torch.autograd.grad(y, [xd] + pd, dy)
        torch.cuda.synchronize()
        out[mode] = (y.detach().cpu().numpy(), [t.cpu().numpy() for t in g])
    monkeypatch.delenv("RD_TILE_WGRAD_GENERIC")
    assert np.array_equal(out["1"][0], out["0"][0])
    differ = 0
    for name, a, r in zip(["x"] + list(ops.ENC_PARAM_NAMES), out["1"][1], out["0"][1]):
        if name == "x" or name.startswith("norm"):
            assert np.array_equal(a, r), name                            # untouched by the switch
        else:
            assert np.abs(a - r).max() <= 2e-5 * np.abs(r).max(), (name, float(np.abs(a - r).max() / np.abs(r).max()))
            differ += int(not np.array_equal(a, r))
    assert differ > 0                                                    # the switch did select another kernel


@pytest.mark.parametrize("T,B,F", [(60, 5, 34), (130, 2, 17), (12, 3, 3)])
def test_materialised_attention_matches_tiled(T, B, F, monkeypatch):
    """The materialised-score attention of wide heads (batched GEMMs around a row softmax; RD_ATTN_BIG=1 forces it at any
    head size) against the flash-style tiled kernels on the same layer WITH dropout: both draw the same Philox quads, so
    output and every gradient agree to the arithmetic's rounding (2e-5 of max-norm in fp32 mode, 1.2e-4 in split-bf16)."""
    from raindrop_amd import _lib, ops
    nhead = 2
    D, nhid = F * 4 + 16, 2 * F * 4
    rng = np.random.default_rng(T + B)
    x = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    lengths = torch.from_numpy(rng.integers(1, T + 1, size=B)).long()
    mask = torch.from_numpy(O2.padding_mask(lengths.numpy(), T)).to(DEV)
    dy = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    p = _enc_params(D, nhid, seed=B)
    shp = _lib.shape(B, T, F, 4, nhead=nhead, nhid=nhid)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RD_ATTN_BIG", mode)
        xd = x.clone().requires_grad_(True)
        pd = [p[n].to(DEV).requires_grad_(True) for n in ops.ENC_PARAM_NAMES]
        y = ops.encoder_layer(xd, mask, shp, 1, 0.25, 99, pd)
        g = torch.autograd.grad(y, [xd] + pd, dy)
        torch.cuda.synchronize()
        out[mode] = (y.detach().cpu().numpy(), [t.cpu().numpy() for t in g])
    monkeypatch.delenv("RD_ATTN_BIG")
    tol = 2e-5 * TOL["x"]
    assert np.abs(out["1"][0] - out["0"][0]).max() <= tol * np.abs(out["0"][0]).max()
    for name, a, r in zip(["x"] + list(ops.ENC_PARAM_NAMES), out["1"][1], out["0"][1]):
        assert np.abs(a - r).max() <= tol * np.abs(r).max(), (name, float(np.abs(a - r).max() / np.abs(r).max()))


def test_encoder_layer_dropout_is_consistent():
    """With dropout on: (i) same seed -> bit-identical output, different seed -> different;
    (ii) backward uses the forward's masks: a central finite difference of <y, R> along a random
    direction of x matches <dx, dir> (the layer is a fixed smooth function once the seed is fixed).
    (The materialised-score attention of wide heads is tied to these kernels, masks included, by
    test_materialised_attention_matches_tiled.)"""
    from raindrop_amd import _lib, ops
    T, B, F, nhead = 12, 2, 3, 2
    D, nhid = F * 4 + 16, 2 * F * 4
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    mask = torch.zeros(B, T, dtype=torch.bool, device=DEV); mask[1, 9:] = True
    p = _enc_params(D, nhid, seed=3)
    pd = [p[n].to(DEV) for n in ops.ENC_PARAM_NAMES]
    shp = _lib.shape(B, T, F, 4, nhead=nhead, nhid=nhid)
    R = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    f = lambda xx, seed=77: ops.encoder_layer(xx, mask, shp, 1, 0.3, seed, pd)
    y1, y2, y3 = f(x), f(x), f(x, 78)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    xg = x.clone().requires_grad_(True)
    (dx,) = torch.autograd.grad((f(xg) * R).sum(), [xg])
    d = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    eps = 2e-2
    fd = (((f(x + eps * d) - f(x - eps * d)) * R).sum() / (2 * eps)).item()
    an = (dx * d).sum().item()
    assert abs(fd - an) < 3e-2 * max(1.0, abs(an)), (fd, an)


def test_dropout_keep_fraction():
    """The Philox masks keep ~ (1-p) of the elements and rescale by 1/(1-p) (observation-embedding site)."""
    from raindrop_amd import _lib, ops
    cfg = synth.make_config("P19")
    B, T, F, d = 16, 60, 34, 4
    K = T * d
    src = torch.ones(T, B, 2 * F, device=DEV)
    shp = _lib.shape(B, T, F, d)
    ident = torch.eye(K, device=DEV)
    zero = torch.zeros(K, device=DEV)
    ones_f = torch.ones(F, device=DEV)
    z, _ = ops.sensor_stage(src, torch.ones(T, B, device=DEV), torch.full((B,), T, device=DEV), ops.timescales(T).to(DEV),
                            ones_f, torch.ones(1, F * d, device=DEV), ident, zero, ident, zero, shp, 0.2, 99)
    v = z[:, :, : F * d]
    kept = (v > 0).float().mean().item()
    assert abs(kept - 0.8) < 0.01
    assert abs(v[v > 0].mean().item() - 1.25) < 1e-5


def test_masked_mean():
    from raindrop_amd import _lib, ops
    T, B, D = 60, 9, 152
    rng = np.random.default_rng(8)
    r = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).requires_grad_(True)
    lengths = torch.from_numpy(rng.integers(1, T + 1, size=B)).long()
    mask = torch.from_numpy(O2.padding_mask(lengths.numpy(), T))
    keep = (~mask).permute(1, 0).unsqueeze(2).float()
    ref = torch.sum(r * keep, dim=0) / (lengths.unsqueeze(1) + 1)
    dout = torch.from_numpy(rng.standard_normal((B, D)).astype(np.float32))
    (g_ref,) = torch.autograd.grad(ref, [r], dout)
    rd = r.detach().to(DEV).requires_grad_(True)
    out = ops.masked_mean(rd, mask.to(DEV), lengths.to(DEV), _lib.shape(B, T, 34, 4))
    (g,) = torch.autograd.grad(out, [rd], dout.to(DEV))
    assert np.abs(out.detach().cpu().numpy() - ref.detach().numpy()).max() < 1e-6
    assert np.abs(g.cpu().numpy() - g_ref.numpy()).max() < 1e-7


def test_flat_adam_matches_torch_adam():
    from raindrop_amd.optim import FlatAdam
    rng = np.random.default_rng(1)
    n = 100003
    p0 = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3)
    mine = torch.nn.Parameter(p0.clone().to(DEV))
    mine.grad = torch.zeros(n, device=DEV)
    fa = FlatAdam(mine, lr=1e-3)
    for s in range(5):
        g = torch.from_numpy(rng.standard_normal(n).astype(np.float32)) * (0.1 + s)
        ref.grad = g.clone(); opt.step()
        mine.grad.copy_(g.to(DEV)); fa.step()
    assert np.abs(mine.detach().cpu().numpy() - ref.detach().numpy()).max() < 2e-6
    # the capturable form (device step cell, bias corrections computed on the device): continues the SAME optimizer state
    dev_p = torch.nn.Parameter(mine.detach().clone()); dev_p.grad = torch.zeros(n, device=DEV)
    fb = FlatAdam(dev_p, lr=1e-3)
    fb.exp_avg.copy_(fa.exp_avg); fb.exp_avg_sq.copy_(fa.exp_avg_sq); fb.t = fa.t
    fb.sync_step_cell()
    for s in range(3):
        g = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(DEV)
        mine.grad.copy_(g); fa.step()
        dev_p.grad.copy_(g); fb.step_captured(); fb.note_replay()
    assert fb.t == fa.t and fb.device_steps() == fa.t
    # (beta^t by repeated multiplication against pow(): the float correction may differ in its last bit -> parameters within an ulp or two)
    a_, b_ = dev_p.detach().cpu().numpy(), mine.detach().cpu().numpy()
    assert np.abs(a_ - b_).max() <= 4e-7 * max(1.0, np.abs(b_).max())


def test_captured_adam_is_race_free_beyond_one_round_of_workgroups():
    """the device-state Adam must not advance {t, beta^t} while workgroups of the same launch can still read
    them (the update launch is read-only on the state now; a launch in front of it advances it).  6 M elements = ~5 900 workgroups,
    several scheduling rounds on 256 CUs (P19's 494 fit in one, which hid the race of round 5's two-slot form): three captured
    steps against three host-state steps, EVERY element; the launches replayed from the same state give the same bits; a
    learning-rate change is a cell update, not a new launch constant."""
    from raindrop_amd.optim import FlatAdam
    n = 6_000_011
    g_ = torch.Generator(device="cpu").manual_seed(5)
    p0 = torch.randn(n, generator=g_)
    host = torch.nn.Parameter(p0.clone().to(DEV)); host.grad = torch.zeros(n, device=DEV)
    devp = torch.nn.Parameter(p0.clone().to(DEV)); devp.grad = torch.zeros(n, device=DEV)
    fa, fb = FlatAdam(host, lr=1e-3), FlatAdam(devp, lr=1e-3)
    fb.sync_step_cell()
    for s in range(3):
        g = (torch.randn(n, generator=g_) * (0.5 + s)).to(DEV)
        if s == 2:
            fa.lr = fb.lr = 2.5e-4
            fb.sync_cell_hyper()
        host.grad.copy_(g); fa.step()
        devp.grad.copy_(g); fb.step_captured(); fb.note_replay()
        torch.cuda.synchronize()
        assert fb.device_steps() == s + 1                                                     # advanced exactly once per step
        d = (devp.detach() - host.detach()).abs().max().item()
        assert d <= 4e-7 * max(1.0, float(host.detach().abs().max())), (s, d)
    # determinism: the same launch from the same state twice
    snap = (devp.detach().clone(), fb.exp_avg.clone(), fb.exp_avg_sq.clone(), fb.step_cell.clone())
    fb.step_captured(); torch.cuda.synchronize()
    first = devp.detach().clone()
    devp.data.copy_(snap[0]); fb.exp_avg.copy_(snap[1]); fb.exp_avg_sq.copy_(snap[2]); fb.step_cell.copy_(snap[3])
    fb.step_captured(); torch.cuda.synchronize()
    assert torch.equal(first, devp.detach())


def test_seed_cell_changes_masks_per_replay():
    """The device seed cell (what a captured hipGraph bumps between replays) is added to every
    by-value dropout seed: same cell value -> identical output, advanced cell -> different masks."""
    import ctypes
    from raindrop_amd import _lib, ops
    T, B, F, nhead = 12, 2, 3, 2
    D, nhid = F * 4 + 16, 2 * F * 4
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.standard_normal((T, B, D)).astype(np.float32)).to(DEV)
    mask = torch.zeros(B, T, dtype=torch.bool, device=DEV)
    pd = [_enc_params(D, nhid, seed=4)[n].to(DEV) for n in ops.ENC_PARAM_NAMES]
    shp = _lib.shape(B, T, F, 4, nhead=nhead, nhid=nhid)
    cell = torch.zeros(1, dtype=torch.int64, device=DEV)
    f = lambda: ops.encoder_layer(x, mask, shp, 0, 0.3, 5, pd)
    base = f()
    try:
        _lib.call("rd_set_seed_cell", ops._ptr(cell))
        y0 = f()
        assert torch.equal(y0, base)                       # cell == 0 is the plain seed
        _lib.call("rd_seed_cell_advance", ops._ptr(cell), 1, ops._stream())
        y1, y1b = f(), f()
        assert not torch.equal(y1, y0) and torch.equal(y1, y1b)
        assert int(cell.item()) == 1
    finally:
        _lib.call("rd_set_seed_cell", None)
    assert torch.equal(f(), base)


# ------------------------------------------------------------------------------------------------
# edge cases and size-independent properties
# ------------------------------------------------------------------------------------------------

def _o2_logits(m, cfg, gs, batch):
    live = set(synth.live_parameter_names(cfg))
    p = {n: t.detach().cpu() for n, t in m.named_parameters() if n in live}
    with torch.no_grad():
        lg, _ = O2.raindrop_v2_forward(p, cfg, batch["src"], batch["static"], batch["times"], batch["lengths"], gs)
    return lg


@pytest.mark.parametrize("case", ["B1", "full_length", "min_length", "no_observations", "self_loops_only"])
def test_model_edge_cases(case):
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "sparse")
    B = 1 if case == "B1" else 5
    batch = synth.make_batch(cfg, B, seed=31)
    T, F = cfg["max_len"], cfg["d_inp"]
    if case == "full_length":            # no padded step at all: mask is all False
        batch["times"] = torch.cumsum(torch.rand(T, B) + 0.01, 0)
    if case == "min_length":             # a single valid step per sample
        batch["times"][1:] = 0
        batch["src"][1:] = 0
    if case == "no_observations":        # every sensor unobserved: X == 0, outputs driven by biases only
        batch["src"].zero_()
    if case == "self_loops_only":        # empty structure: only the diagonal the model adds itself
        gs = torch.zeros(F, F)
    batch["lengths"] = torch.sum(batch["times"] > 0, dim=0)
    m = build_ours(cfg, gs, DEV, 13).eval()
    dv = {k: (None if v is None else v.to(DEV)) for k, v in batch.items()}
    with torch.no_grad():
        logits, dist, _ = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
    ref = _o2_logits(m, cfg, gs, batch)
    assert torch.isfinite(logits).all() and float(dist) == 0.0
    assert np.abs(logits.cpu().numpy() - ref.numpy()).max() < 1e-4


def test_batch_invariance_at_validation_scale():
    """`evaluate_standard` (code/utils_rd.py:310-320) pushes the whole validation split through ONE
    forward.  Every stage is per-sample, so a 1500-sample forward must equal the concatenation of
    its chunks bit-for-bit in the fused path's arithmetic (same kernels, same per-sample order)."""
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "ones")
    N = 1500
    batch = synth.make_batch(cfg, N, seed=77)
    m = build_ours(cfg, gs, DEV, 21).eval()
    dv = {k: (None if v is None else v.to(DEV)) for k, v in batch.items()}
    with torch.no_grad():
        full, _, _ = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
        parts = []
        for lo in range(0, N, 500):
            hi = lo + 500
            lg, _, _ = m(dv["src"][:, lo:hi].contiguous(), dv["static"][lo:hi].contiguous(),
                         dv["times"][:, lo:hi].contiguous(), dv["lengths"][lo:hi].contiguous())
            parts.append(lg)
    assert full.shape == (N, 2) and torch.isfinite(full).all()
    assert torch.equal(full, torch.cat(parts, 0))
    ref = _o2_logits(m, cfg, gs, {k: (v[:, :8] if k in ("src", "times") else (v[:8] if v is not None else None))
                                  for k, v in batch.items()})
    assert np.abs(full[:8].cpu().numpy() - ref.numpy()).max() < 1e-4


def test_setting3_sensor_removal_is_exact_zeroing():
    """Setting 3 (code/Raindrop.py:216-226) zeroes random sensor columns of val/test samples; the
    model must treat a zeroed sensor exactly like an unobserved one (X == 0 -> same logits as the
    oracle on the same zeroed input)."""
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "ones")
    batch = synth.make_batch(cfg, 6, seed=5)
    rng = np.random.default_rng(0)
    F = cfg["d_inp"]
    for i in range(6):
        idx = rng.choice(F, round(0.5 * F), replace=False)
        batch["src"][:, i, idx] = 0
    m = build_ours(cfg, gs, DEV, 3).eval()
    dv = {k: (None if v is None else v.to(DEV)) for k, v in batch.items()}
    with torch.no_grad():
        logits, _, _ = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
    assert np.abs(logits.cpu().numpy() - _o2_logits(m, cfg, gs, batch).numpy()).max() < 1e-4


def test_empty_batch_is_a_no_op():
    from raindrop_amd import _lib, ops
    cfg = synth.make_config("TINY")
    m = build_ours(cfg, synth.make_structure(cfg, "ones"), DEV, 1).eval()
    T, F = cfg["max_len"], cfg["d_inp"]
    with torch.no_grad():
        logits, _, _ = m(torch.zeros(T, 0, 2 * F, device=DEV), torch.zeros(0, cfg["d_static"], device=DEV),
                         torch.zeros(T, 0, device=DEV), torch.zeros(0, dtype=torch.int64, device=DEV))
    assert logits.shape == (0, cfg["n_classes"])


@pytest.mark.parametrize("T,B,D,Fe,ds,C", [(60, 37, 152, 34, 6, 2), (7, 3, 36, 5, 3, 2), (50, 5, 84, 0, 0, 8), (30, 4, 160, 36, 9, 2),
                                           (12, 1, 16, 4, 2, 3)])
def test_fused_head_against_float64(T, B, D, Fe, ds, C):
    """rd_head_train (masked mean -> [agg | emb] -> mlp_static -> mean cross entropy, and every gradient, two launches)
    against the same chain in float64 torch autograd (code/models_rd.py:366-385, code/Raindrop.py:255,322).  The kernel is
    fp32 FMA in every precision mode: 2e-5 of each tensor's max-norm (1e-6 on the loss)."""
    from raindrop_amd import _lib
    lib = _lib.load()
    assert lib.rd_head_train_supported(D, Fe, C)
    rng = np.random.default_rng(T * 100 + B)
    f = lambda *sh: torch.from_numpy(rng.standard_normal(sh).astype(np.float32))
    dh = D + Fe
    r = f(T, B, D)
    lengths = torch.from_numpy(rng.integers(1, T + 1, size=B)).long()
    mask = torch.from_numpy(O2.padding_mask(lengths.numpy(), T))
    stat = f(B, ds) if Fe else None
    par = {"emb_w": f(Fe, ds) * 0.3 if Fe else None, "emb_b": f(Fe) * 0.1 if Fe else None, "w0": f(dh, dh) * 0.1, "b0": f(dh) * 0.1,
           "w2": f(C, dh) * 0.2, "b2": f(C) * 0.1}
    y = torch.from_numpy(rng.integers(0, C, size=B)).long()
    # ---- float64 reference ----
    r64 = r.double().requires_grad_(True)
    p64 = {k: (None if v is None else v.double().requires_grad_(True)) for k, v in par.items()}
    keep = (~mask).double().t().unsqueeze(-1)                                   # [T,B,1]
    agg = (r64 * keep).sum(0) / (lengths.double() + 1).unsqueeze(-1)
    feat = torch.cat([agg, stat.double() @ p64["emb_w"].t() + p64["emb_b"]], 1) if Fe else agg
    logits64 = torch.relu(feat @ p64["w0"].t() + p64["b0"]) @ p64["w2"].t() + p64["b2"]
    loss64 = torch.nn.functional.cross_entropy(logits64, y)
    names = [k for k in ("emb_w", "emb_b", "w0", "b0", "w2", "b2") if par[k] is not None]
    gref = torch.autograd.grad(loss64, [r64] + [p64[k] for k in names])
    # ---- device ----
    dev = lambda t: None if t is None else t.to(DEV).contiguous()
    rd, md, ld, sd, yd = dev(r), dev(mask), dev(lengths), dev(stat), dev(y)
    pd = {k: dev(v) for k, v in par.items()}
    gd = {k: (None if v is None else torch.full_like(pd[k], 7.0)) for k, v in par.items()}
    loss = torch.zeros((), device=DEV); logits = torch.empty(B, C, device=DEV); dr = torch.full((T, B, D), 7.0, device=DEV)
    ws = torch.empty(lib.rd_head_train_workspace_bytes(B, dh, C), dtype=torch.uint8, device=DEV)
    P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    shp = _lib.shape(B, T, 1, 4)
    _lib.call("rd_head_train", shp, D, ds, Fe, C, P(rd), P(md), P(ld), P(sd), P(pd["emb_w"]), P(pd["emb_b"]), P(pd["w0"]), P(pd["b0"]),
              P(pd["w2"]), P(pd["b2"]), P(yd), P(loss), P(logits), P(gd["emb_w"]), P(gd["emb_b"]), P(gd["w0"]), P(gd["b0"]),
              P(gd["w2"]), P(gd["b2"]), P(dr), P(ws), ws.numel(), None)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss64.detach())) < 1e-6 * max(1.0, abs(float(loss64.detach())))
    assert np.abs(logits.cpu().numpy() - logits64.detach().numpy()).max() < 2e-5 * np.abs(logits64.detach().numpy()).max()
    for name, got, ref in zip(["r"] + names, [dr] + [gd[k] for k in names], gref):
        a, b = got.cpu().numpy().astype(np.float64), ref.numpy()
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), (name, float(np.abs(a - b).max() / np.abs(b).max()))


@pytest.mark.parametrize("use_graph,fused_head", [(False, True), (True, True), (True, False)])
def test_static_train_step_matches_autograd(use_graph, fused_head, precision_mode, monkeypatch):
    """raindrop_amd.step.TrainStep (explicit fwd+CE+bwd, optionally one hipGraph) against the autograd path: loss and
    every gradient must agree to rounding.  With the classifier head operator by operator (RD_HEAD_FUSED=0) the step
    calls the very kernels autograd calls: 1e-5.  The fused head (rd_head.hip) computes in fp32 FMA, autograd's head in
    the mode's arithmetic: the split-bf16 products differ from fp32 by a few 1e-6 per term, hence 5e-5 there."""
    from raindrop_amd import dp
    from raindrop_amd.step import TrainStep
    monkeypatch.setenv("RD_HEAD_FUSED", "1" if fused_head else "0")
    tol = 5e-5 if (fused_head and precision_mode != "fp32") else 1e-5
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "sparse")
    batch = synth.make_batch(cfg, 8, seed=41)
    dv = {k: (None if v is None else v.to(DEV)) for k, v in batch.items()}
    m = build_ours(cfg, gs, DEV, 7).train()                     # dropout p forced to 0 by build_ours
    live = synth.live_parameter_names(cfg)
    named = dict(m.named_parameters())
    logits, _, _ = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
    loss = torch.nn.functional.cross_entropy(logits, dv["y"])
    ref = torch.autograd.grad(loss, [named[n] for n in live])
    flat = dp.FlatGradAllReduce([(n, named[n]) for n in live])
    step = TrainStep(m, flat, dv, use_graph=use_graph)
    try:
        for _ in range(2):                                       # replays must be idempotent without dropout
            l2 = step.run()
        torch.cuda.synchronize()
        assert step.head_fused == fused_head
        assert abs(float(l2) - float(loss)) < (1e-6 if tol == 1e-5 else 5e-6)
        for n, g in zip(live, ref):
            got = named[n].grad
            assert _rel(got.cpu().numpy(), g.cpu().numpy()) < tol, n
    finally:
        step.close()


def test_split_k_weight_gradients_vs_float64():
    """rd_linear_bwd_weight (split over the M rows, fixed-order reduce; bias gradient riding on the same product) on the
    layer shapes of the path and ragged row counts, against float64."""
    from raindrop_amd import ops
    rng = np.random.default_rng(5)
    for (M, N, K) in [(15360, 152, 272), (5000, 272, 152), (4099, 152, 152), (3001, 456, 152), (8704, 240, 240), (1025, 64, 48)]:
        x = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).to(DEV).requires_grad_(True)
        W = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).to(DEV).requires_grad_(True)
        bb = torch.zeros(N, device=DEV, requires_grad=True)
        dy = torch.from_numpy(rng.standard_normal((M, N)).astype(np.float32)).to(DEV)
        _, hW, hb = torch.autograd.grad(ops.linear(x, W, bb, 0), [x, W, bb], dy)
        refW = (dy.double().t() @ x.detach().double()).cpu().numpy()
        refb = dy.double().sum(0).cpu().numpy()
        assert _rel(hW.cpu().numpy(), refW) < 2e-5 * TOL["x"], (M, N, K)
        assert _rel(hb.cpu().numpy(), refb) < 2e-5, (M, N, K)


def test_static_train_step_dropout_varies_per_replay():
    """Under graph replay the by-value seed is frozen; the device seed cell, bumped by the graph itself,
    must give every replay fresh dropout masks (different loss), deterministically (same sequence twice)."""
    from raindrop_amd import dp
    from raindrop_amd.step import TrainStep
    cfg = synth.make_config("P19")
    batch = synth.make_batch(cfg, 8, seed=43)
    dv = {k: (None if v is None else v.to(DEV)) for k, v in batch.items()}
    seqs = []
    for _ in range(2):
        m = build_ours(cfg, synth.make_structure(cfg, "ones"), DEV, 7).train()
        m.dropout.p = 0.2
        named = dict(m.named_parameters())
        flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)])
        step = TrainStep(m, flat, dv, use_graph=True)
        try:
            seqs.append([float(step.run()) for _ in range(4)])
        finally:
            step.close()
    assert len(set(seqs[0])) == 4                 # four replays, four different masks
    assert seqs[0] == seqs[1]                     # and the sequence is reproducible


@pytest.mark.parametrize("F,T,B,kind", [(34, 60, 9, "sparse"), (34, 60, 256, "ones"), (34, 60, 100, "sparse"), (16, 60, 3, "ones"),
                                        (17, 32, 5, "sparse"), (48, 60, 4, "ones"), (64, 60, 3, "sparse"), (33, 16, 7, "ones")])
def test_k1_fused_vs_generic_same_arithmetic(F, T, B, kind, precision_mode, monkeypatch):
    """The fused K1 kernels (rd_msgpass_fused.hip + rd_msgpass_dw.hip: row tiles, bit-mask gates, streamed dW)
    against the generic tiled-GEMM path IN THE SAME split-bf16 ARITHMETIC.  Both evaluate relu(x W^T + b) with
    the same 3-product split and fp32 accumulation and differ only in summation order (~1e-6), so the bound is
    tight: forward 5e-6 of max-norm; every gradient entry within 2e-5 of the tensor's max-norm, except the rows a
    ReLU gate flip can move (a pre-activation within ~1e-6 of zero opening on one path only): at most 0.1 % of the
    entries may exceed the bound and the relative L2 error stays below 1e-3.  A dropped or mis-indexed term -- the
    two leftover graph rows of a sample, a missing dW slice, a wrong gate bit -- fails these by orders of magnitude."""
    if precision_mode != "bf16x3":
        pytest.skip("the fused path exists in split-bf16 mode only")
    from raindrop_amd import _lib, ops
    d, K = 4, T * 4
    rng = np.random.default_rng(F * 1000 + T * 10 + B)
    cfgF = dict(d_inp=F, max_len=T, static=True, d_static=3, n_classes=2)
    b = synth.make_batch(cfgF, B, seed=F + B, density=0.5)
    gs = torch.ones(F, F) if kind == "ones" else synth.make_structure(dict(d_inp=F), "sparse")
    names = ["R_u", "W1", "b1", "W2", "b2"]
    shapes = [(1, F * d), (K, K), (K,), (K, K), (K,)]
    p = {n: synth.param_values("k1." + n, s, seed=3) for n, s in zip(names, shapes)}
    p["R_u"] = p["R_u"] * 3.0
    dz = torch.from_numpy(rng.standard_normal((T, B, F * d + 16)).astype(np.float32)).to(DEV)
    adj, _, _ = ops.graph_build(gs.to(DEV))
    _, ssum = ops.edge_softmax_dense(adj)
    shp = _lib.shape(B, T, F, d)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RD_K1_FUSED", mode)
        pd = {n: t.detach().to(DEV).requires_grad_(True) for n, t in p.items()}
        z, mask = ops.sensor_stage(b["src"].to(DEV), b["times"].to(DEV), b["lengths"].to(DEV), ops.timescales(T).to(DEV),
                                   ssum, pd["R_u"], pd["W1"], pd["b1"], pd["W2"], pd["b2"], shp)
        g = torch.autograd.grad(z, [pd[n] for n in names], dz)
        torch.cuda.synchronize()
        out[mode] = (z.detach().cpu().numpy(), mask.cpu().numpy(), [x.cpu().numpy() for x in g])
    monkeypatch.delenv("RD_K1_FUSED")
    zf, mf, gf = out["1"]
    zg, mg, gg = out["0"]
    assert np.array_equal(mf, mg)
    assert np.array_equal(zf[:, :, F * d:], zg[:, :, F * d:])                     # PE columns: same arithmetic, bit-equal
    assert np.abs(zf - zg).max() <= 5e-6 * np.abs(zg).max(), np.abs(zf - zg).max() / np.abs(zg).max()
    for n, a, r in zip(names, gf, gg):
        scale = np.abs(r).max() + 1e-30
        bad = np.abs(a - r) > 2e-5 * scale
        assert bad.mean() <= 1e-3, (n, float(bad.mean()), float(np.abs(a - r).max() / scale))
        assert _rel2(a, r) < 1e-3, (n, _rel2(a, r))


@pytest.mark.parametrize("F,T,B", [(36, 215, 100), (18, 130, 120)])
def test_unfused_k1_streamed_weight_gradients_match_split_k(F, T, B, precision_mode, monkeypatch):
    """The unfused message passing at shapes with >= 2048 graph rows and K >= 512 (P12: 36 sensors x 215 steps): its two [K, K] weight
    gradients through the conversion pass + tile stream (round 4) against the split-K GEMM pair (RD_K1_WGRAD_STREAM=0), same
    split-bf16 arithmetic and dropout masks: identical output and R_u gradient, dW / db within 2e-5 of the tensor's max-norm.
    B*F = 3600 / 2160 rows (>= 4 K) end in a partial 32-row chunk; K = 860 / 520 is not a multiple of the 64-column conversion blocks."""
    if precision_mode != "bf16x3":
        pytest.skip("the tile stream exists in the bf16 modes only")
    from raindrop_amd import _lib, ops
    d = 4
    K = T * d
    rng = np.random.default_rng(F * T)
    b = synth.make_batch(dict(d_inp=F, max_len=T, static=True, d_static=3, n_classes=2), B, seed=B)
    names = ["R_u", "W1", "b1", "W2", "b2"]
    shapes = [(1, F * d), (K, K), (K,), (K, K), (K,)]
    p = {n: synth.param_values("k1w." + n, s, seed=9).to(DEV) for n, s in zip(names, shapes)}
    adj, _, _ = ops.graph_build(torch.ones(F, F, device=DEV))
    _, ssum = ops.edge_softmax_dense(adj)
    shp = _lib.shape(B, T, F, d)
    dz = torch.from_numpy(rng.standard_normal((T, B, F * d + 16)).astype(np.float32)).to(DEV)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RD_K1_WGRAD_STREAM", mode)
        q = {n: t.clone().requires_grad_(True) for n, t in p.items()}
        z, mask = ops.sensor_stage(b["src"].to(DEV), b["times"].to(DEV), b["lengths"].to(DEV), ops.timescales(T).to(DEV), ssum,
                                   q["R_u"], q["W1"], q["b1"], q["W2"], q["b2"], shp, 0.2, 11)
        g = torch.autograd.grad(z, [q[n] for n in names], dz)
        torch.cuda.synchronize()
        res[mode] = [z.detach().cpu().numpy()] + [x.cpu().numpy() for x in g]
    monkeypatch.delenv("RD_K1_WGRAD_STREAM")
    assert np.array_equal(res["1"][0], res["0"][0]) and np.array_equal(res["1"][1], res["0"][1])      # z, dR_u: untouched by the switch
    differ = 0
    for n, a, r in zip(names[1:], res["1"][2:], res["0"][2:]):
        assert np.abs(a - r).max() <= 2e-5 * np.abs(r).max(), (n, float(np.abs(a - r).max() / np.abs(r).max()))
        differ += int(not np.array_equal(a, r))
    assert differ > 0                                                    # the switch did select another kernel


@pytest.mark.parametrize("B,p_drop", [(7, 0.0), (256, 0.2)])
def test_k1_shape_specialised_kernels_equal_generic_instantiation(B, p_drop, precision_mode, monkeypatch):
    """The P19 shape runs instantiations of the fused kernels with F = 34, T = 60 as compile-time constants (index arithmetic
    folded; rd_msgpass_fused.hip `Dim`).  Same source, same arithmetic, same order: output, mask and all five gradients must be
    BIT-equal to the runtime-shape instantiation (RD_K1_SPECIALIZE=0)."""
    if precision_mode != "bf16x3":
        pytest.skip("the fused path exists in split-bf16 mode only")
    from raindrop_amd import _lib, ops
    F, T, d = 34, 60, 4
    K = T * d
    rng = np.random.default_rng(B)
    b = synth.make_batch(dict(d_inp=F, max_len=T, static=True, d_static=3, n_classes=2), B, seed=B)
    names = ["R_u", "W1", "b1", "W2", "b2"]
    shapes = [(1, F * d), (K, K), (K,), (K, K), (K,)]
    p = {n: synth.param_values("k1s." + n, s, seed=5).to(DEV) for n, s in zip(names, shapes)}
    adj, _, _ = ops.graph_build(torch.ones(F, F, device=DEV))
    _, ssum = ops.edge_softmax_dense(adj)
    shp = _lib.shape(B, T, F, d)
    dz = torch.from_numpy(rng.standard_normal((T, B, F * d + 16)).astype(np.float32)).to(DEV)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RD_K1_SPECIALIZE", mode)
        q = {n: t.clone().requires_grad_(True) for n, t in p.items()}
        z, mask = ops.sensor_stage(b["src"].to(DEV), b["times"].to(DEV), b["lengths"].to(DEV), ops.timescales(T).to(DEV), ssum,
                                   q["R_u"], q["W1"], q["b1"], q["W2"], q["b2"], shp, p_drop, 11)
        g = torch.autograd.grad(z, [q[n] for n in names], dz)
        torch.cuda.synchronize()
        res[mode] = [z.detach().cpu().numpy(), mask.cpu().numpy()] + [x.cpu().numpy() for x in g]
    monkeypatch.delenv("RD_K1_SPECIALIZE")
    for n, a, r in zip(["z", "mask"] + names, res["1"], res["0"]):
        assert np.array_equal(a, r), n


def test_k1_fused_dropout_backward_uses_forward_mask(precision_mode, monkeypatch):
    """Dropout on the observation embedding, fused path: the forward pass hands its keep mask to the backward pass as
    gate bits (no regeneration).  Same seed -> identical output, another seed -> another mask; the generic path draws
    the SAME Philox mask (same (seed, site, cell) function), so output and all five gradients must agree with it as
    tightly as without dropout; and dR_u matches a central finite difference of <z, R> (once the seed is fixed the
    stage is a fixed piecewise-linear function of R_u with few kinks near the sample)."""
    if precision_mode != "bf16x3":
        pytest.skip("the fused path exists in split-bf16 mode only")
    from raindrop_amd import _lib, ops
    F, T, B, d = 34, 60, 5, 4
    K = T * d
    rng = np.random.default_rng(77)
    b = synth.make_batch(dict(d_inp=F, max_len=T, static=True, d_static=3, n_classes=2), B, seed=5, density=0.6)
    names = ["R_u", "W1", "b1", "W2", "b2"]
    shapes = [(1, F * d), (K, K), (K,), (K, K), (K,)]
    p = {n: synth.param_values("k1d." + n, s, seed=4).to(DEV) for n, s in zip(names, shapes)}
    p["R_u"] = p["R_u"] * 3.0
    adj, _, _ = ops.graph_build(torch.ones(F, F, device=DEV))
    _, ssum = ops.edge_softmax_dense(adj)
    shp = _lib.shape(B, T, F, d)
    R = torch.from_numpy(rng.standard_normal((T, B, F * d + 16)).astype(np.float32)).to(DEV)
    src, times, lengths, ts = b["src"].to(DEV), b["times"].to(DEV), b["lengths"].to(DEV), ops.timescales(T).to(DEV)

    def f(q, seed=9):
        z, _ = ops.sensor_stage(src, times, lengths, ts, ssum, q["R_u"], q["W1"], q["b1"], q["W2"], q["b2"], shp, 0.3, seed)
        return z
    z1, z2, z3 = f(p), f(p), f(p, 10)
    assert torch.equal(z1, z2) and not torch.equal(z1, z3)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RD_K1_FUSED", mode)
        q = {n: t.clone().requires_grad_(True) for n, t in p.items()}
        z = f(q)
        g = torch.autograd.grad((z * R).sum(), [q[n] for n in names])
        res[mode] = (z.detach().cpu().numpy(), [x.cpu().numpy() for x in g])
    monkeypatch.delenv("RD_K1_FUSED")
    (zf, gf), (zg, gg) = res["1"], res["0"]
    assert np.abs(zf - zg).max() <= 5e-6 * np.abs(zg).max()
    for n, a, r in zip(names, gf, gg):
        scale = np.abs(r).max() + 1e-30
        assert (np.abs(a - r) > 2e-5 * scale).mean() <= 1e-3 and _rel2(a, r) < 1e-3, (n, _rel2(a, r))
    eps = 1e-2
    dirn = torch.from_numpy(rng.standard_normal(tuple(p["R_u"].shape)).astype(np.float32)).to(DEV)
    pp = dict(p); pm = dict(p)
    pp["R_u"] = p["R_u"] + eps * dirn; pm["R_u"] = p["R_u"] - eps * dirn
    fd = (((f(pp) - f(pm)) * R).sum() / (2 * eps)).item()
    an = float((torch.from_numpy(gf[0]).to(DEV) * dirn).sum())
    assert abs(fd - an) < 3e-2 * max(1.0, abs(an)), (fd, an)


@pytest.mark.parametrize("name", ["p12_b32", "p19_b256"])
def test_plain_bf16_mode_against_golden(name):
    """RD_PREC_BF16 (one bf16 product per MFMA step: BASELINE.json's 'P12 ... bf16' arithmetic) against the reference's
    fp32 fixtures at the looser bound SURVEY 8c states for it: logits within 3e-2, loss within 1e-2, gradients within 15 %
    relative L2 (no reference bf16 path exists to compare bit patterns with)."""
    from raindrop_amd import _lib
    g, meta = load_golden(name)
    cfg, gs, batch = case_inputs(meta)
    _lib.call("rd_set_precision", 2)
    try:
        m = build_ours(cfg, gs, DEV, meta["param_seed"]).train()
        dv = {k: (None if v is None else v.to(DEV)) for k, v in batch.items()}
        logits, _, _ = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
        loss = torch.nn.functional.cross_entropy(logits, dv["y"])
        loss.backward()
        assert np.abs(logits.detach().cpu().numpy() - g["logits"]).max() < 3e-2
        assert abs(loss.item() - float(g["loss"])) < 1e-2
        params = dict(m.named_parameters())
        for n in (str(x) for x in g["live"]):
            exp, got = golden_grad(g, n, params[n].grad)
            assert _rel2(got, exp) < 0.15, (n, _rel2(got, exp))      # R_u (the deepest, smallest gradient) measures 6 %
    finally:
        _lib.call("rd_set_precision", 1)


def test_edge_coefficient_dropout_of_the_operator():
    """code/Ob_propagation.py:196: `gamma = F.dropout(gamma, p, training)` AFTER the softmax (the shipped model builds the operator with
    dropout 0; refused until round 6).  RNG streams cannot match torch's, so the test pins the semantics: every dropped-and-rescaled
    coefficient is 0 or gamma / (1 - p), the kept fraction is ~1 - p, the output is relu(lin_value(x)) times the per-target SUM of
    those coefficients, a fixed seed reproduces the mask, eval mode and p = 0 are the plain softmax."""
    from raindrop_amd import ops
    from raindrop_amd.Ob_propagation import Observation_progation
    rng = np.random.default_rng(3)
    n, K, p = 34, 240, 0.3
    adj = (rng.random((n, n)) < 0.6).astype(np.float32) * rng.uniform(0.5, 1.5, (n, n)).astype(np.float32)
    ei_np, ew_np = O2.build_graph(adj)
    ei, ew = torch.from_numpy(ei_np).to(DEV), torch.from_numpy(ew_np).to(DEV)
    g0, s0 = ops.edge_softmax_list(ei, ew, n, norm_row=1)
    g1, s1 = ops.edge_softmax_list(ei, ew, n, norm_row=1, p_drop=p, seed=11)
    g2, s2 = ops.edge_softmax_list(ei, ew, n, norm_row=1, p_drop=p, seed=11)
    g3, _ = ops.edge_softmax_list(ei, ew, n, norm_row=1, p_drop=p, seed=12)
    assert torch.equal(g1, g2) and torch.equal(s1, s2) and not torch.equal(g1, g3)
    kept = g1 != 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.05
    assert float((g1[kept] - g0[kept] / (1 - p)).abs().max()) < 1e-6
    ref_sum = torch.zeros(n, device=DEV).index_add_(0, ei[1], g1)
    assert float((s1 - ref_sum).abs().max()) < 1e-5
    op = Observation_progation(K, K, n_nodes=n, ob_dim=4, heads=1, dropout=p)
    synth.fill_params_(op, seed=4)
    op = op.to(DEV).train()
    x = torch.from_numpy(rng.standard_normal((n, K)).astype(np.float32)).to(DEV)
    torch.manual_seed(5); op._drop_calls = 0
    y1 = op(x, p_t=None, edge_index=ei, edge_weights=ew)
    torch.manual_seed(5); op._drop_calls = 0
    y2 = op(x, p_t=None, edge_index=ei, edge_weights=ew)
    y3 = op(x, p_t=None, edge_index=ei, edge_weights=ew)                     # next call: another mask
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    v = torch.relu(torch.nn.functional.linear(x, op.lin_value.weight, op.lin_value.bias))
    ratio = (y1 / v.clamp_min(1e-20))[v > 1e-3].detach()                           # = the target's dropped coefficient sum, per row
    assert float(ratio.min()) >= -1e-4 and float(ratio.max()) <= 1.0 / (1 - p) + 1e-3
    op.eval()
    ye = op(x, p_t=None, edge_index=ei, edge_weights=ew)
    assert float((ye - v * s0[:, None]).abs().max()) <= 2e-4 * float(v.abs().max())


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_transformer_conv_general_form_vs_reference_fixture(tag):
    """Round 6 (VERDICT r5 missing #4): `TransformerConv` WITHOUT given edge weights -- q.k scores per edge and head, `lin_edge` on
    the key, concat / head mean, root weight with and without the beta gate -- against the reference's own class
    (tests/golden/tconv_general.npz, make_goldens.py TCONV_GENERAL; a 9-node graph with two duplicated edges): output, the returned
    post-softmax coefficients, gradients w.r.t. x, edge_attr and every parameter.  Training-mode coefficient dropout: determinism,
    kept fraction, and the eval path unchanged."""
    import os
    from raindrop_amd.transformer_conv import TransformerConv
    from tests.helpers import GOLDEN
    from tests.golden.make_goldens import TCONV_GENERAL
    g = np.load(os.path.join(GOLDEN, "tconv_general.npz"))
    _, H, C, concat, edim, beta, root, pseed = next(c for c in TCONV_GENERAL if c[0] == tag)
    tc = TransformerConv(7, C, heads=H, concat=concat, beta=beta, edge_dim=edim, root_weight=root)
    synth.fill_params_(tc, seed=pseed)
    tc = tc.to(DEV).train()
    x = torch.from_numpy(g["x"]).to(DEV).requires_grad_(True)
    ei = torch.from_numpy(g["ei"]).to(DEV)
    ea = torch.from_numpy(g[tag + "_ea"]).to(DEV).requires_grad_(True) if edim is not None else None
    y, (ei_o, alpha) = tc(x, ei, edge_weights=None, edge_attr=ea, return_attention_weights=True)
    tol = 2e-5 * TOL["x"]
    assert torch.equal(ei_o, ei)
    assert np.abs(alpha.detach().cpu().numpy() - g[tag + "_alpha"]).max() < tol
    assert _rel(y.detach().cpu().numpy(), g[tag + "_y"]) < tol
    names = [k for k, _ in tc.named_parameters()]
    wrt = [x] + ([ea] if ea is not None else []) + [p_ for _, p_ in tc.named_parameters()]
    grads = torch.autograd.grad((y * torch.from_numpy(g[tag + "_R"]).to(DEV)).sum(), wrt, allow_unused=True)
    assert _rel(grads[0].cpu().numpy(), g[tag + "_gx"]) < 5 * tol
    gi = 1
    if ea is not None:
        assert _rel(grads[1].cpu().numpy(), g[tag + "_gea"]) < 5 * tol
        gi = 2
    for k, got in zip(names, grads[gi:]):
        key = tag + "_g/" + k
        if key not in g.files:
            continue
        if k == "lin_key.bias":                  # the softmax is shift-invariant per target: analytically zero, rounding noise on both sides
            assert got is not None and float(got.abs().max()) < 1e-5 and np.abs(g[key]).max() < 1e-5
        else:
            assert got is not None and _rel(got.cpu().numpy(), g[key]) < 5 * tol, k
    # coefficient dropout (code/transformer_conv.py:203): masks are a function of (torch seed, call counter)
    tc.dropout = 0.4
    torch.manual_seed(3); tc._drop_calls = 0
    y1 = tc(x, ei, edge_attr=ea)
    torch.manual_seed(3); tc._drop_calls = 0
    y2 = tc(x, ei, edge_attr=ea)
    assert torch.equal(y1, y2) and not torch.equal(y1, y)
    tc.eval()
    ye = tc(x, ei, edge_attr=ea)
    assert float((ye - y).abs().max()) <= 1e-6 * max(1.0, float(y.abs().max()))


def test_transformer_conv_general_form_edge_cases():
    """The general form on the lists the fixture does not hold: an EMPTY edge list (out = root weight only, alpha [0, H]), nodes
    without in-edges (their aggregate is zero) next to a node with 300 in-edges (more than one pass of the 256-thread scan), and an
    endpoint out of range (IndexError, as the reference's index_select raises).  Checked against the same sums written with torch
    scatter operations in fp64."""
    from raindrop_amd.transformer_conv import TransformerConv
    N, H, C = 11, 3, 5
    tc = TransformerConv(7, C, heads=H, concat=True, root_weight=True)
    synth.fill_params_(tc, seed=5)
    tc = tc.to(DEV).eval()
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(N, 7, generator=gen).to(DEV).requires_grad_(True)
    # (1) empty list
    e0 = torch.zeros((2, 0), dtype=torch.int64, device=DEV)
    y0, (_, a0) = tc(x, e0, return_attention_weights=True)
    skip = torch.nn.functional.linear(x.double(), tc.lin_skip.weight.double(), tc.lin_skip.bias.double())
    assert tuple(a0.shape) == (0, H)
    assert float((y0.double() - skip).abs().max()) < 2e-5 * TOL["x"] * float(skip.abs().max())
    g0, = torch.autograd.grad(y0.sum(), x)
    assert float((g0.double() - tc.lin_skip.weight.double().sum(0)[None, :]).abs().max()) < 1e-4
    # (2) 300 edges into node 4, a few into node 9, none into the others (duplicates included)
    src = torch.randint(0, N, (310,), generator=gen)
    tgt = torch.cat([torch.full((300,), 4), torch.full((10,), 9)])
    ei = torch.stack([src, tgt]).to(DEV)
    y, (_, al) = tc(x, ei, return_attention_weights=True)
    xd = x.double()
    q = torch.nn.functional.linear(xd, tc.lin_query.weight.double(), tc.lin_query.bias.double()).view(N, H, C)
    k = torch.nn.functional.linear(xd, tc.lin_key.weight.double(), tc.lin_key.bias.double()).view(N, H, C)
    v = torch.nn.functional.linear(xd, tc.lin_value.weight.double(), tc.lin_value.bias.double()).view(N, H, C)
    s = (q[ei[1]] * k[ei[0]]).sum(-1) / C ** 0.5                                  # [E, H]
    ref_al = torch.zeros_like(s)
    for t in (4, 9):
        m = ei[1] == t
        ref_al[m] = torch.softmax(s[m], dim=0)
    agg = torch.zeros(N, H, C, dtype=torch.float64, device=DEV).index_add_(0, ei[1], ref_al[:, :, None] * v[ei[0]])
    ref = agg.view(N, H * C) + skip
    assert float((al.double() - ref_al).abs().max()) < 2e-5 * TOL["x"]
    assert float((y.double() - ref).abs().max()) < 2e-5 * TOL["x"] * float(ref.abs().max())
    R = torch.randn(N, H * C, generator=gen).to(DEV)
    gx, = torch.autograd.grad((y * R).sum(), x)
    gref, = torch.autograd.grad((ref * R.double()).sum(), x)
    assert float((gx.double() - gref).abs().max()) < 1e-4 * TOL["x"] * float(gref.abs().max())
    # (3) an endpoint out of range
    bad = ei.clone(); bad[0, 7] = N
    with pytest.raises(IndexError):
        tc(x, bad)
    bad = ei.clone(); bad[1, 3] = -1
    with pytest.raises(IndexError):
        tc(x, bad)
