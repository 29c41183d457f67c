"""CPU: the restatement (oracle O2) must reproduce every committed golden vector, which were
produced by the reference's own files (oracle O1, tests/golden/make_goldens.py)."""
import numpy as np
import pytest
import torch

from oracle import restatement as O2
from raindrop_amd import synth
from tests.helpers import BETA_CASES, MODEL_CASES, case_inputs, load_golden

FAST = [c for c in MODEL_CASES if c != "pam_ones"]   # PAM (150 M parameters) runs in the slow set


def _params(cfg, gs, meta, live):
    """Reference-shaped parameter dict without building any model: names/shapes from the
    committed state_dict surface, values from the seeded fill (dead parameters never enter the
    forward and are skipped)."""
    import json
    import os
    from tests.helpers import GOLDEN
    surf = json.load(open(os.path.join(GOLDEN, "state_dict_surface.json")))[meta["cfg"]]
    return {k: synth.param_values(k, surf[k], meta["param_seed"], float(meta.get("param_scale", 1.0))).requires_grad_(True)
            for k in sorted(surf) if k in live}


@pytest.mark.parametrize("name", FAST + ["pam_ones"])
def test_restatement_matches_golden(name):
    g, meta = load_golden(name)
    cfg, gs, batch = case_inputs(meta)
    live = set(str(x) for x in g["live"])
    p = _params(cfg, gs, meta, live)
    logits, loss, grads = O2.step_fwd_bwd(p, cfg, batch, gs, faithful=False)
    assert np.abs(logits.numpy() - g["logits"]).max() < 2e-6
    assert abs(float(loss) - float(g["loss"])) < 2e-6
    assert float(g["distance"]) == 0.0
    for n in live:
        exp = g["grad/" + n]
        got = grads[n].reshape(-1)[:: int(g["gradstride/" + n])].numpy()
        scale = np.abs(exp).max() + 1e-30
        assert np.abs(got - exp).max() / scale < 5e-5, n
        assert abs(grads[n].double().norm().item() - float(g["gradnorm/" + n])) <= 1e-4 * float(g["gradnorm/" + n]) + 1e-12


@pytest.mark.parametrize("name", BETA_CASES)
def test_restatement_matches_use_beta_golden(name):
    """The paper's branch: the reference's forward with `use_beta = True` (code/models_rd.py:317 flipped in memory when the
    fixture was made) against the restatement's use_beta order -- logits, loss, the structure distance (non-zero here: the
    samples prune different edges) and every gradient, including the three tensors only this branch trains."""
    g, meta = load_golden(name)
    cfg, gs, batch = case_inputs(meta)
    live = set(str(x) for x in g["live"])
    assert {"ob_propagation.map_weights", "ob_propagation.increase_dim.weight", "ob_propagation.increase_dim.bias"} <= live
    p = _params(cfg, gs, meta, live)
    logits, loss, grads = O2.step_fwd_bwd(p, cfg, batch, gs, faithful=True, use_beta=True)
    with torch.no_grad():
        _, dist = O2.raindrop_v2_forward(p, cfg, batch["src"], batch["static"], batch["times"], batch["lengths"], gs,
                                         faithful=True, use_beta=True)
    assert np.abs(logits.numpy() - g["logits"]).max() < 2e-6
    assert abs(float(loss) - float(g["loss"])) < 2e-6
    assert float(g["distance"]) > 0.0 and abs(float(dist) - float(g["distance"])) < 1e-6
    for n in live:
        exp = g["grad/" + n]
        got = grads[n].reshape(-1)[:: int(g["gradstride/" + n])].numpy()
        scale = np.abs(exp).max() + 1e-30
        assert np.abs(got - exp).max() / scale < 2e-4, n            # tied scores: summation order of the kept edges differs
        assert abs(grads[n].double().norm().item() - float(g["gradnorm/" + n])) <= 1e-4 * float(g["gradnorm/" + n]) + 1e-12


@pytest.mark.parametrize("name", ["tiny_sparse", "p19_sparse"])
def test_faithful_order_matches_golden(name):
    """The per-sample / per-edge evaluation order (what the reference literally does) agrees too."""
    g, meta = load_golden(name)
    cfg, gs, batch = case_inputs(meta)
    p = _params(cfg, gs, meta, set(str(x) for x in g["live"]))
    with torch.no_grad():
        logits, dist = O2.raindrop_v2_forward(p, cfg, batch["src"], batch["static"], batch["times"],
                                              batch["lengths"], gs, faithful=True)
    assert np.abs(logits.numpy() - g["logits"]).max() < 2e-6
    assert float(dist) == 0.0


@pytest.mark.parametrize("name", MODEL_CASES)
def test_integer_work_bit_exact(name):
    g, meta = load_golden(name)
    cfg, gs, batch = case_inputs(meta)
    ei, ew = O2.build_graph(gs.numpy())
    assert np.array_equal(ei, g["edge_index"]) and np.array_equal(ew, g["edge_weights"])
    assert np.array_equal(O2.padding_mask(batch["lengths"].numpy(), cfg["max_len"]), g["mask"])
    assert np.array_equal(O2.lengths_from_times(batch["times"].numpy()), g["lengths"])


def test_operator_goldens():
    g = np.load(__import__("os").path.join(__import__("tests.helpers", fromlist=["GOLDEN"]).GOLDEN, "operators.npz"))
    t = torch.from_numpy
    ei, ew = O2.build_graph(g["obp_adj"])
    y, a = O2.observation_propagation(t(g["obp_x"]), t(ei), t(ew), t(g["obp_w"]), t(g["obp_b"]))
    assert np.abs(y.numpy() - g["obp_y"]).max() < 1e-6
    assert np.array_equal(a.numpy(), g["obp_alpha"]) and np.array_equal(ei, g["obp_ei"])
    yb, (eib, ab) = O2.observation_propagation_beta(
        t(g["obp_x"]), t(g["obp_p_t"]), t(ei), t(ew), t(g["obp_w"]), t(g["obp_b"]),
        t(g["obpb_w_inc"]), t(g["obpb_b_inc"]), t(g["obpb_map"]), 4)
    assert np.abs(yb.numpy() - g["obpb_y"]).max() < 1e-6
    assert np.array_equal(eib.numpy(), g["obpb_ei"])
    assert np.abs(ab.numpy() - g["obpb_alpha"]).max() < 1e-7
    ei2, ew2 = O2.build_graph(g["tc_adj"])
    y2, a2 = O2.transformer_conv(t(g["tc_x"]), t(ei2), t(ew2), t(g["tc_wv"]), t(g["tc_bv"]),
                                 t(g["tc_ws"]), t(g["tc_bs"]))
    assert np.abs(y2.numpy() - g["tc_y"]).max() < 1e-6
    assert np.abs(a2.numpy() - g["tc_alpha"]).max() < 1e-7


def test_large_graph_beta_golden():
    """O2's use_beta operator against the reference's outputs on the 256-node, 13 k-edge fixture (tests/golden/beta_large.npz:
    the graph size the workspace form of rd_graph_beta_fwd exists for); scores are distinct by construction of the fixture, so
    the pruned edge list is reproducible bit for bit."""
    import os
    from raindrop_amd import synth
    from raindrop_amd.Ob_propagation import Observation_progation
    from tests.helpers import GOLDEN
    g = np.load(os.path.join(GOLDEN, "beta_large.npz"))
    n, T, d, B = (int(v) for v in g["dims"])
    K = T * d
    op = Observation_progation(K, K, n_nodes=n, ob_dim=d, heads=1)
    synth.fill_params_(op, seed=31)
    t = torch.from_numpy
    ei, ew = O2.build_graph(g["adj"])
    assert ei.shape[1] > 8192 and n > 64                                         # beyond the LDS-staged kernels' envelope
    for b in range(B):
        y, (ei2, a2) = O2.observation_propagation_beta(t(g["X"][b]), t(g["PT"][b]), t(ei), t(ew), op.lin_value.weight.detach(),
                                                       op.lin_value.bias.detach(), op.increase_dim.weight.detach(),
                                                       op.increase_dim.bias.detach(), op.map_weights.detach(), d)
        assert np.array_equal(ei2.numpy(), g["ei"][b])
        assert np.abs(a2.numpy().ravel() - g["alpha"][b].ravel()).max() < 1e-7
        assert np.abs(y.numpy() - g["Y"][b]).max() < 1e-5


def test_restatement_in_float64_bounds_the_fp32_reference_error():
    """How far the fp32 CPU reference order itself is from exact arithmetic (the yardstick for the GPU ties in
    tests/test_token_plan_gpu.py::test_benchmarked_step_against_float64): the restatement on the same fp32 parameters and inputs
    in float64 vs fp32 -- logits within 1e-6, every gradient within 5e-6 in relative L2 at P19, B = 37."""
    from tests.helpers import build_ours
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "sparse")
    batch = synth.make_batch(cfg, 37, seed=3)
    live = set(synth.live_parameter_names(cfg))
    m = build_ours(cfg, gs, "cpu", 7)
    p32 = {n: t.detach().clone().requires_grad_(True) for n, t in m.named_parameters() if n in live}
    p64 = {n: t.detach().double().requires_grad_(True) for n, t in m.named_parameters() if n in live}
    b64 = {k: (v.double() if (v is not None and v.is_floating_point()) else v) for k, v in batch.items()}
    lg32, ls32, gr32 = O2.step_fwd_bwd(p32, cfg, batch, gs, faithful=False)
    lg64, ls64, gr64 = O2.step_fwd_bwd(p64, cfg, b64, gs.double(), faithful=False)
    assert lg64.dtype == torch.float64
    assert float((lg32.double() - lg64).abs().max()) < 1e-6
    assert abs(float(ls32) - float(ls64)) < 1e-6
    for n in gr64:
        assert float((gr32[n].double() - gr64[n]).norm() / (gr64[n].norm() + 1e-300)) < 5e-6, n


def test_restatement_of_the_general_transformer_conv_matches_the_fixture():
    """O2's `transformer_conv_general` against the reference's own TransformerConv without edge weights (tconv_general.npz)."""
    import os
    from tests.helpers import GOLDEN
    from tests.golden.make_goldens import TCONV_GENERAL
    from raindrop_amd.transformer_conv import TransformerConv
    g = np.load(os.path.join(GOLDEN, "tconv_general.npz"))
    for tag, H, C, concat, edim, beta, root, pseed in TCONV_GENERAL:
        tc = TransformerConv(7, C, heads=H, concat=concat, beta=beta, edge_dim=edim, root_weight=root)      # parameter names / shapes only
        synth.fill_params_(tc, seed=pseed)
        p = {k: v.detach() for k, v in tc.named_parameters()}
        ea = torch.from_numpy(g[tag + "_ea"]) if edim is not None else None
        y, a = O2.transformer_conv_general(torch.from_numpy(g["x"]), torch.from_numpy(g["ei"]), p, H, C, concat, ea, root)
        assert np.abs(y.numpy() - g[tag + "_y"]).max() < 2e-6 and np.abs(a.numpy() - g[tag + "_alpha"]).max() < 1e-6, tag
