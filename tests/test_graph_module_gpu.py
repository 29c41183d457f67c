"""GPU: the captured step behind the nn.Module surface (raindrop_amd/graph_module.py; the DEFAULT since round 5): what the reference's
unchanged loop -- model.forward, criterion, loss.backward(), torch's optimizer (code/Raindrop.py:310-324) -- gets, against the
eager operator-by-operator autograd path of the same module."""
import numpy as np
import pytest
import torch

from raindrop_amd import _lib, synth
from tests.helpers import build_ours

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _close(got, ref):
    """to rounding (5e-5 of the max-norm) -- or, where a ReLU gate whose pre-activation is within ~1e-6 of zero opens on one path
    only (the two paths sum in different orders), a few rows off by a visible amount: relative L2 below 2e-3 (DESIGN (c))"""
    a, b = got.cpu().numpy().astype(np.float64), ref.cpu().numpy().astype(np.float64)
    return _rel(a, b) < 5e-5 or float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)) < 2e-3


def _batch(cfg, B, seed):
    return {k: (None if v is None else v.to(DEV)) for k, v in synth.make_batch(cfg, B, seed=seed).items()}


def _loop_step(m, dv, graph):
    """one iteration of the reference's loop body on module m; returns (logits, loss, {name: grad})"""
    m.graph_step = graph
    for p in m.parameters():
        p.grad = None
    logits, dist, _ = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
    loss = torch.nn.functional.cross_entropy(logits, dv["y"])
    loss.backward()
    return logits.detach().clone(), float(loss), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}, dist


@pytest.mark.parametrize("cfg_name,B", [("P19", 8), ("P19", 37), ("P12", 4), ("PAM", 2)])
def test_module_graph_step_matches_eager_autograd(cfg_name, B):
    """Dropout off: logits, loss and every live gradient of the two-graph module step equal the eager autograd path's (the
    captured step runs the token plan and the fused kernels where the shape has them, the eager path the padded layout: the same
    function of the inputs; fused head in fp32 FMA against the mode's products: 5e-5).  Replays on a SECOND batch (the graphs
    were captured on the first) and after an optimizer step (parameters change in place) must follow."""
    cfg = synth.make_config(cfg_name)
    gs = synth.make_structure(cfg, "sparse")
    m = build_ours(cfg, gs, DEV, 7).train()                     # dropout p forced to 0 by build_ours
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    live = set(synth.live_parameter_names(cfg))
    for it, seed in enumerate((41, 42, 43)):
        dv = _batch(cfg, B, seed)
        lg_e, loss_e, g_e, _ = _loop_step(m, dv, False)
        lg_g, loss_g, g_g, dist = _loop_step(m, dv, True)
        assert float(dist) == 0.0
        assert set(g_g) == set(g_e) == live, (sorted(set(g_g) ^ live), sorted(set(g_e) ^ live))
        assert np.abs(lg_g.cpu().numpy() - lg_e.cpu().numpy()).max() < 2e-5, it
        assert abs(loss_g - loss_e) < 5e-6, it
        for n in sorted(live):
            assert _close(g_g[n], g_e[n]), (it, n, _rel(g_g[n].cpu().numpy(), g_e[n].cpu().numpy()))
        opt.step()                                              # torch's own Adam on p.grad (the graph path's, here)
    assert len(m._graph_runners) == 1                           # one capture served all three batches


def test_module_graph_step_dropout_and_fallbacks():
    """Dropout on: fresh masks per call (the forward graph bumps the seed cell), reproducible across identical models; evaluation and
    no-grad calls, a model with use_beta, and a stale backward do what the module docstring says."""
    cfg = synth.make_config("P19")
    dv = _batch(cfg, 16, 5)
    seqs = []
    for _ in range(2):
        m = build_ours(cfg, synth.make_structure(cfg, "ones"), DEV, 7).train()
        m.dropout.p = 0.2
        seqs.append([_loop_step(m, dv, True)[1] for _ in range(3)])
    assert len(set(seqs[0])) == 3 and seqs[0] == seqs[1]
    # evaluation / no-grad calls stay on the eager path and do not disturb the runner
    m.eval()
    with torch.no_grad():
        lg_eval = m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0]
    m.graph_step = False
    with torch.no_grad():
        lg_ref = m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0]
    assert torch.equal(lg_eval, lg_ref)
    # two forwards before a backward: the second call takes the eager path (the first one's activations stay in the runner), both
    # backwards work and give what two eager calls give
    m.train(); m.graph_step = True; m.dropout.p = 0.0
    for p in m.parameters():
        p.grad = None
    l1 = m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0]
    l2 = m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0]
    assert sum(bool(r and r.busy()) for r in m._graph_runners.values()) == 1      # the runner of this dropout setting holds l1's activations
    l1.sum().backward()
    l2.sum().backward()
    both = m.mlp_static[2].weight.grad.clone()
    m.graph_step = False
    for p in m.parameters():
        p.grad = None
    m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0].sum().backward()
    assert torch.allclose(both, 2 * m.mlp_static[2].weight.grad, rtol=1e-3, atol=1e-6)   # one captured + one eager call against two eager ones
    # a captured forward whose output was dropped without a backward does not block the next one
    m.graph_step = True
    m(dv["src"], dv["static"], dv["times"], dv["lengths"])
    assert not any(r.busy() for r in m._graph_runners.values() if r)
    # a stale autograd node (forced: the runner's generation moved on) refuses instead of returning wrong gradients
    l1 = m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0]
    next(r for r in m._graph_runners.values() if r and r.busy()).gen += 1
    with pytest.raises(_lib.RaindropHipError):
        l1.sum().backward()
    m.dropout.p = 0.2
    # gradient accumulation over two forward/backward pairs without zero_grad: p.grad is the sum (autograd accumulates copies)
    m.dropout.p = 0.0
    for p in m.parameters():
        p.grad = None
    for _ in range(2):
        m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0].square().sum().backward()
    twice = m.mlp_static[2].weight.grad.clone()
    for p in m.parameters():
        p.grad = None
    m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0].square().sum().backward()
    assert torch.allclose(twice, 2 * m.mlp_static[2].weight.grad, rtol=1e-6, atol=0)
    # the paper's branch is not covered by the captured step: eager path -- BIT-equal to the switch being off --, gradients for increase_dim
    mb = build_ours(cfg, synth.make_structure(cfg, "sparse"), DEV, 7, use_beta=True).train()
    res = []
    for flag in (True, False, None):                            # None: the default
        mb.graph_step = flag
        for p in mb.parameters():
            p.grad = None
        lg = mb(dv["src"], dv["static"], dv["times"], dv["lengths"])[0]
        lg.sum().backward()
        res.append((lg.detach().clone(), mb.ob_propagation.increase_dim.weight.grad.clone(), mb.R_u.grad.clone()))
    assert not getattr(mb, "_graph_runners", {})
    for r in res[1:]:
        assert all(torch.equal(x, y) for x, y in zip(res[0], r))
    # evaluation with gradients enabled (model.eval(), no torch.no_grad()): eager path, bit-equal to the switch being off
    m.eval(); m.graph_step = None
    lg_a = m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0]
    m.graph_step = False
    lg_b = m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0]
    assert torch.equal(lg_a, lg_b)


def test_module_graph_step_environment_switch(monkeypatch):
    """An unchanged script (no attribute on the model, nothing in the environment) gets the captured step; RD_MODULE_GRAPH=0
    switches it off."""
    cfg = synth.make_config("TINY")
    dv = _batch(cfg, 3, 9)
    m = build_ours(cfg, synth.make_structure(cfg, "ones"), DEV, 7).train()
    monkeypatch.delenv("RD_MODULE_GRAPH", raising=False)
    assert getattr(m, "graph_step", None) is None
    lg = m(dv["src"], dv["static"], dv["times"], dv["lengths"])[0]
    assert len(getattr(m, "_graph_runners", {})) == 1 and lg.grad_fn is not None
    monkeypatch.setenv("RD_MODULE_GRAPH", "0")
    m2 = build_ours(cfg, synth.make_structure(cfg, "ones"), DEV, 7).train()
    lg2 = m2(dv["src"], dv["static"], dv["times"], dv["lengths"])[0]
    assert not getattr(m2, "_graph_runners", {})
    assert np.abs(lg.detach().cpu().numpy() - lg2.detach().cpu().numpy()).max() < 2e-5


def test_module_graph_step_recaptures_when_it_must():
    """A second batch size gets its own captured pair; a parameter whose storage was replaced (the graphs hold addresses) makes
    the runner stale: the next call captures again instead of training a dangling copy."""
    cfg = synth.make_config("P19")
    m = build_ours(cfg, synth.make_structure(cfg, "ones"), DEV, 7).train()
    live = sorted(synth.live_parameter_names(cfg))
    for B in (8, 5, 8):
        dv = _batch(cfg, B, 60 + B)
        _, loss_e, g_e, _ = _loop_step(m, dv, False)
        _, loss_g, g_g, _ = _loop_step(m, dv, True)
        assert abs(loss_g - loss_e) < 5e-6
        assert all(_close(g_g[n], g_e[n]) for n in live)
    assert len(m._graph_runners) == 2
    old = m._graph_runners[next(k for k in m._graph_runners if k[1] == 8)]
    w = m.mlp_static[0].weight
    w.data = (w.data * 1.5).clone()                              # new storage, new values
    dv = _batch(cfg, 8, 77)
    _, loss_e, g_e, _ = _loop_step(m, dv, False)
    _, loss_g, g_g, _ = _loop_step(m, dv, True)
    assert m._graph_runners[next(k for k in m._graph_runners if k[1] == 8)] is not old
    assert abs(loss_g - loss_e) < 5e-6 and all(_close(g_g[n], g_e[n]) for n in live)


def test_module_graph_step_keeps_a_bounded_number_of_runners(monkeypatch):
    """Each runner holds a whole step's activations: a loop that varies its batch size keeps the RD_MODULE_GRAPH_MAX most recently
    created ones (the oldest idle one goes), and the results stay those of the eager path."""
    monkeypatch.setenv("RD_MODULE_GRAPH_MAX", "2")
    cfg = synth.make_config("P19")
    m = build_ours(cfg, synth.make_structure(cfg, "ones"), DEV, 9).train()
    live = sorted(synth.live_parameter_names(cfg))
    for B in (4, 6, 8, 4):
        dv = _batch(cfg, B, 80 + B)
        _, loss_e, g_e, _ = _loop_step(m, dv, False)
        _, loss_g, g_g, _ = _loop_step(m, dv, True)
        assert abs(loss_g - loss_e) < 5e-6 and all(_close(g_g[n], g_e[n]) for n in live)
        assert len(m._graph_runners) <= 2
    assert sorted(k[1] for k in m._graph_runners) == [4, 8]      # 6 was the oldest when 4 came back


@pytest.mark.parametrize("views", ["1", "0"], ids=["grad_views", "returned_clones"])
def test_module_graph_step_keeps_autograd_s_accumulation_semantics(views, monkeypatch):
    """Round 6: the captured backward node SETS each p.grad to a persistent view of its flat gradient buffer instead of returning 35
    fresh views to autograd (RD_MODULE_GRAD_VIEWS=0: the round-5 form).  Whatever the form, the loop-visible semantics are
    autograd's: (a) a second forward + backward without zero_grad adds to the gradients that are there, (b) zero_grad(set_to_none=
    False) zeroes in place and the next backward starts from zero, (c) a gradient another loss term left in p.grad is added to."""
    monkeypatch.setenv("RD_MODULE_GRAD_VIEWS", views)
    cfg = synth.make_config("P19")
    m = build_ours(cfg, synth.make_structure(cfg, "ones"), DEV, 7).train()
    live = sorted(synth.live_parameter_names(cfg))
    named = dict(m.named_parameters())
    A, Bb = _batch(cfg, 16, 51), _batch(cfg, 16, 52)
    _, _, gA, _ = _loop_step(m, A, False)
    _, _, gB, _ = _loop_step(m, Bb, False)

    def fb(dv):
        m.graph_step = True
        logits, _, _ = m(dv["src"], dv["static"], dv["times"], dv["lengths"])
        torch.nn.functional.cross_entropy(logits, dv["y"]).backward()

    for p in m.parameters():
        p.grad = None
    fb(A); fb(Bb)                                                  # (a) no zero_grad in between
    for n in live:
        assert _close(named[n].grad, gA[n] + gB[n]), ("accumulate", n)
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    opt.zero_grad(set_to_none=False)                               # (b) zeroed in place
    fb(A)
    for n in live:
        assert _close(named[n].grad, gA[n]), ("zero_grad in place", n)
    opt.zero_grad(set_to_none=True)
    n0 = "mlp_static.2.weight"
    named[n0].grad = torch.ones_like(named[n0])                    # (c) a foreign gradient
    fb(Bb)
    assert _close(named[n0].grad, gB[n0] + 1.0)
    for n in live:
        if n != n0:
            assert _close(named[n].grad, gB[n]), ("fresh", n)


def test_autograd_step_captures_the_paper_s_branch():
    """`AutogradStep`: Raindrop_v2(use_beta=True, compute_distance=True) -- which `TrainStep` refuses -- as one hipGraph per step over
    the module's autograd surface.  Dropout off: three replays on three batches (copied into the static buffers) give the eager loop's
    losses, logits and gradients bit for bit (same kernels, same order); p19_beta_sparse's golden logits within 1e-4."""
    from raindrop_amd.step import AutogradStep
    from tests.helpers import load_golden, case_inputs
    g, meta = load_golden("p19_beta_sparse")
    cfg, gs, batch = case_inputs(meta)
    outs = []
    for graph in (True, False):
        m = build_ours(cfg, gs, DEV, meta["param_seed"], use_beta=True, compute_distance=True).train()
        m.graph_step = False
        buf = {k: (None if v is None else v.to(DEV).clone()) for k, v in batch.items()}
        st = AutogradStep(m, buf, optimizer=False) if graph else None
        res = []
        for seed in (meta["batch_seed"], 77, 78):
            nb = synth.make_batch(cfg, meta["batch"], seed=seed)
            for k, v in nb.items():
                if v is not None:
                    buf[k].copy_(v)
            if graph:
                loss = float(st.run()); lg = st.logits.clone()
            else:
                for p in m.parameters():
                    p.grad = None
                lg, _, _ = m(buf["src"], buf["static"], buf["times"], buf["lengths"])
                l_ = torch.nn.functional.cross_entropy(lg, buf["y"]); l_.backward(); loss = float(l_); lg = lg.detach().clone()
            res.append((loss, lg, {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
        outs.append(res)
    assert np.abs(outs[0][0][1].cpu().numpy() - g["logits"]).max() < 1e-4
    for (l0, lg0, g0), (l1, lg1, g1) in zip(*outs):
        assert l0 == l1 and torch.equal(lg0, lg1) and set(g0) == set(g1)
        for n in g0:
            assert torch.equal(g0[n], g1[n]), n
