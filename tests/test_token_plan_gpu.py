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
    coff = np.concatenate([[0], np.cumsum((lenr + 15) // 16)])          # 16-row groups of the per-sample group space
    return dict(off=off, rank=rank, order=order, lenr=lenr, cnt=cnt, coff=coff, mlive=int(off[-1]))


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
    assert np.array_equal(p[o:o + T + 1], ref["cnt"]); o += T + 1
    assert np.array_equal(p[o:o + B + 1], ref["coff"]) and p[5] == (ref["coff"][-1] + 1) // 2; o += B + 1
    assert np.array_equal(p[o:o + B], ref["off"][ref["rank"]]); o += B              # brow[b]: first row of sample b
    assert np.array_equal(p[o:o + B], np.clip(lengths, 0, T))                      # blen[b]


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


# ------------------------------------------------------------------------------------------------------------------------------
# One capture, many batches: the real use of the step (code/Raindrop.py:310-324 draws a new batch every iteration; INTEGRATION.md
# "static training step": `DeviceDataset.batch(idx, out=batch); step.run()`).  New `lengths` change the plan INSIDE the captured
# graph (rd_step_begin): live-row count, rank order, block height and input slack all move, and rows beyond the new M_live still
# hold the previous batch's values.  Every result must equal a FRESH step built on that batch alone.
# ------------------------------------------------------------------------------------------------------------------------------
REPLAY_SEQ = ["random", "full_length", "short", "min_length", "first_time_zero", "one_long", "random"]


def _seq_batch(cfg, B, case, seed):
    if case == "short":                                         # fewer live rows than "random": lengths ~ U[2, T/4]
        batch = synth.make_batch(cfg, B, seed=seed + 1)
        T = cfg["max_len"]
        cut = torch.from_numpy(np.random.default_rng(seed).integers(2, T // 4 + 1, size=B))
        dead = torch.arange(T)[:, None] >= cut[None, :]
        batch["times"][dead] = 0
        batch["src"][dead] = 0
        batch["lengths"] = torch.sum(batch["times"] > 0, dim=0)
        return batch
    return _case_batch(cfg, B, case, seed)


def _make_step(cfg, gs, dv, token_plan, use_graph, p_drop, split=False, seed_params=7):
    from raindrop_amd.step import TrainStep
    m = build_ours(cfg, gs, DEV, seed_params).train()
    m.dropout.p = p_drop
    live = synth.live_parameter_names(cfg)
    named = dict(m.named_parameters())
    flat = dp.FlatGradAllReduce([(n, named[n]) for n in live])
    step = TrainStep(m, flat, dv, use_graph=use_graph, token_plan=token_plan, autotune=False, split=split)
    return step, named, live


def _snapshot(step, named, live):
    torch.cuda.synchronize()
    return (float(step.loss), step.logits.cpu().numpy().copy(), {n: named[n].grad.detach().cpu().numpy().copy() for n in live})


def _load_into(dv, batch):
    for k, v in batch.items():
        if v is not None:
            dv[k].copy_(v.to(dv[k].dtype))


@pytest.mark.parametrize("split", [False, True])
def test_captured_plan_graph_replayed_on_new_batches(split):
    """Capture ONCE on batch A, then copy batches with more / fewer / minimal live rows, a slack-1 batch and A again into the SAME
    buffers and replay.  Reference: a fresh padded-layout TrainStep on each batch (dropout off): loss 2e-6, logits 2e-6,
    every gradient 2e-5 of its max-norm.  `split`: the two-graph data-parallel form (graph A | graph B)."""
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "sparse")
    B = 256
    batches = [_seq_batch(cfg, B, c, seed=61 + 3 * i) for i, c in enumerate(REPLAY_SEQ)]
    batches[-1] = batches[0]                                                # the first batch again at the end
    dv = {k: (None if v is None else v.to(DEV).clone()) for k, v in batches[0].items()}
    step, named, live = _make_step(cfg, gs, dv, True, True, 0.0, split=split)
    assert step.plan is not None and step.graph is not None and (step.graph_b is not None) == split
    got, mlives = [], []
    for bt in batches:
        _load_into(dv, bt)
        step.run()
        got.append(_snapshot(step, named, live))
        mlives.append(int(step.plan[0]))
    step.close()
    T = cfg["max_len"]
    for bt, ml in zip(batches, mlives):
        assert ml == int(torch.clamp(bt["lengths"], 0, T).sum())
    assert mlives[1] > mlives[0] > mlives[2] > mlives[3]                   # more, fewer, minimal live rows than the capture batch
    assert got[-1][0] == got[0][0] and np.array_equal(got[-1][1], got[0][1])      # A again: the same bits as the first time
    for n in live:
        assert np.array_equal(got[-1][2][n], got[0][2][n]), n
    for i, bt in enumerate(batches[:-1]):
        dref = {k: (None if v is None else v.to(DEV)) for k, v in bt.items()}
        ref, rnamed, _ = _make_step(cfg, gs, dref, False, False, 0.0)
        ref.run()
        rl, rg, rgr = _snapshot(ref, rnamed, live)
        ref.close()
        l, g, gr = got[i]
        assert abs(l - rl) < 2e-6 * max(1.0, abs(rl)), (REPLAY_SEQ[i], l, rl)
        assert np.abs(g - rg).max() < 2e-6, (REPLAY_SEQ[i], float(np.abs(g - rg).max()))
        for n in live:
            assert _rel(gr[n], rgr[n]) < 2e-5, (REPLAY_SEQ[i], n, _rel(gr[n], rgr[n]))


@pytest.mark.parametrize("split", [False, True])
def test_captured_plan_graph_replay_with_dropout_equals_fresh_capture(split):
    """Dropout ON: the masks are functions of (seed, seed cell, compact row), so a replay on batch k with the cell at value c must
    give the SAME BITS as a fresh token-plan step whose first run on batch k happens with the cell at c -- whatever the replayed
    step's buffers held before (rows beyond the new M_live keep the previous batch's values)."""
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "ones")
    B = 128
    seq = ["full_length", "random", "short", "min_length", "first_time_zero", "random"]
    batches = [_seq_batch(cfg, B, c, seed=17 + 5 * i) for i, c in enumerate(seq)]
    dv = {k: (None if v is None else v.to(DEV).clone()) for k, v in batches[0].items()}
    step, named, live = _make_step(cfg, gs, dv, True, True, 0.2, split=split)
    got = []
    for k, bt in enumerate(batches):
        _load_into(dv, bt)
        step.seed_cell.fill_(1000 + k)                                      # the graph bumps it by one before its first kernel
        step.run()
        got.append(_snapshot(step, named, live))
    step.close()
    assert len({g[0] for g in got}) == len(got)
    for k, bt in enumerate(batches):
        dref = {kk: (None if v is None else v.to(DEV)) for kk, v in bt.items()}
        ref, rnamed, _ = _make_step(cfg, gs, dref, True, False, 0.2)      # eager token-plan step on fresh (zeroed) buffers
        ref.seed_cell.fill_(1000 + k)
        ref.run()
        rl, rg, rgr = _snapshot(ref, rnamed, live)
        ref.close()
        l, g, gr = got[k]
        assert np.isfinite(l) and l == rl, (seq[k], l, rl)
        assert np.array_equal(g, rg), seq[k]
        for n in live:
            assert np.array_equal(gr[n], rgr[n]), (seq[k], n, _rel(gr[n], rgr[n]))


POISON_GROUPS = ["z", "x", "dx", "enc_saved", "enc_ws", "k1_saved", "k1_ws", "head"]


@pytest.mark.parametrize("group", POISON_GROUPS)
def test_token_plan_step_ignores_stale_buffer_contents(group):
    """A diverged step can leave inf / NaN in rows that a later, shorter batch does not rewrite (rows at or beyond M_live; the
    padded layout rewrote every row every step).  Fill each scratch / saved buffer group of the step with NaN bit patterns
    (fp32 NaN = bf16 NaN pair) between two runs on the same batch: the second run must give the same bits as the first."""
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "ones")
    batch = _seq_batch(cfg, 96, "random", seed=5)
    dv = {k: (None if v is None else v.to(DEV).clone()) for k, v in batch.items()}
    step, named, live = _make_step(cfg, gs, dv, True, True, 0.0)
    step.run()
    first = _snapshot(step, named, live)
    bufs = {"z": [step.z], "x": step.x[1:], "dx": step.dx, "enc_saved": step.enc_saved, "enc_ws": [step.enc_ws],
            "k1_saved": [step.k1_saved], "k1_ws": [step.k1_ws],
            "head": [t for t in (step.head_ws, step.feat, step.dfeat, step.hid, step.dhid) if t is not None]}[group]
    for t in bufs:
        t.view(torch.uint8).reshape(-1)[: t.numel() * t.element_size() // 4 * 4].view(torch.int32).fill_(0x7FC07FC0)
    step.run()
    second = _snapshot(step, named, live)
    step.close()
    assert np.isfinite(second[0]) and second[0] == first[0], (group, first[0], second[0])
    assert np.array_equal(second[1], first[1]), group
    for n in live:
        assert np.array_equal(second[2][n], first[2][n]), (group, n)


@pytest.mark.parametrize("p_drop", [0.0, 0.2])
@pytest.mark.parametrize("case", ["random", "full_length", "min_length", "first_time_zero"])
def test_fused_attention_launch_vs_the_three_launches_it_replaces(case, p_drop, monkeypatch):
    """rd_attnfuse.hip (in_proj + attention + in_proj's input gradient as one launch per direction, row tiles exported in the
    per-sample chunk space) against the round-3 path (RD_ATTN_FUSE=0: QKV row-block product, attention per (sample, head), QKV
    input-gradient product) on the same token plan, same seed cell -- dropout masks are identical functions of (seed, rank, head,
    query, key), so the two agree to the order of a few fp32 sums: loss 2e-6, logits 2e-6, every gradient 2e-5 of its max-norm."""
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "sparse")
    batch = _seq_batch(cfg, 64, case, seed=77)
    res = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("RD_ATTN_FUSE", fuse)
        dv = {k: (None if v is None else v.to(DEV).clone()) for k, v in batch.items()}
        step, named, live = _make_step(cfg, gs, dv, True, False, p_drop)
        step.seed_cell.fill_(5)
        step.run()
        res.append(_snapshot(step, named, live))
        step.close()
    (l1, g1, gr1), (l0, g0, gr0) = res
    assert np.isfinite(l1) and abs(l1 - l0) < 2e-6 * max(1.0, abs(l0)), (l1, l0)
    assert np.abs(g1 - g0).max() < 2e-6, float(np.abs(g1 - g0).max())
    for n in live:
        assert _rel(gr1[n], gr0[n]) < 2e-5, (n, _rel(gr1[n], gr0[n]))


# ------------------------------------------------------------------------------------------------------------------------------
# Round 4: the plan beyond the P19 envelope -- P12 (T = 215: the panel-product message passing whose last scatter follows the plan,
# the multi-tile attention kernels on plan rows, the D = 160 / nhid = 288 chains) and the single-product (bf16) arithmetic mode.
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg_name,B,mode", [("P12", 12, 1), ("P12", 40, 1), ("P19", 48, 2), ("P12", 12, 2)])
@pytest.mark.parametrize("case", ["random", "min_length", "first_time_zero"])
def test_token_plan_beyond_the_p19_envelope(cfg_name, B, mode, case):
    """TrainStep on the compact layout against TrainStep on the padded layout, same batch, same parameters, dropout off, in the
    arithmetic mode `mode` (1 = bf16x3, 2 = bf16 single product: compared with ITSELF on the padded layout, so the bounds stay
    the summation-order ones -- loss 2e-6, logits 2e-6 (2e-5 in the single-product mode, whose per-sample arithmetic follows the
    row tiling), gradients 2e-5 of max-norm (1e-3 in the single-product mode: 8-bit mantissas amplify every reordering)."""
    cfg = synth.make_config(cfg_name)
    gs = synth.make_structure(cfg, "sparse")
    batch = _case_batch(cfg, B, case, seed=91)
    _lib.call("rd_set_precision", mode)
    try:
        la, ga, gra, sa = _run_step(cfg, gs, batch, True, True, steps=2)
        lb, gb, grb, sb_ = _run_step(cfg, gs, batch, False, True, steps=2)
    finally:
        _lib.call("rd_set_precision", 1)
    ltol, gtol = (2e-6, 2e-5) if mode == 1 else (2e-5, 1e-3)
    assert la[0] == la[1]
    assert abs(la[1] - lb[1]) < ltol * max(1.0, abs(lb[1])), (la, lb)
    assert np.abs(ga - gb).max() < ltol, float(np.abs(ga - gb).max())
    for n in gra:
        assert _rel(gra[n], grb[n]) < gtol, (n, _rel(gra[n], grb[n]))
    p = sa.plan.cpu().numpy()
    assert p[0] == int(torch.clamp(batch["lengths"], 0, cfg["max_len"]).sum())
    sa.close(); sb_.close()


def test_p12_plan_graph_replayed_on_new_batches_with_dropout():
    """The P12 step captured once and replayed on batches of other lengths, dropout on: bit-equal to a fresh eager step on the same
    plan and seed cell (the multi-tile attention and the plan-following scatter see rows of the previous batch beyond M_live)."""
    cfg = synth.make_config("P12")
    gs = synth.make_structure(cfg, "ones")
    B = 24
    seq = ["full_length", "random", "short", "random"]
    batches = [_seq_batch(cfg, B, c, seed=27 + 5 * i) for i, c in enumerate(seq)]
    dv = {k: (None if v is None else v.to(DEV).clone()) for k, v in batches[0].items()}
    step, named, live = _make_step(cfg, gs, dv, True, True, 0.2)
    assert step.plan is not None
    got = []
    for k, bt in enumerate(batches):
        _load_into(dv, bt)
        step.seed_cell.fill_(500 + k)
        step.run()
        got.append(_snapshot(step, named, live))
    step.close()
    for k, bt in enumerate(batches):
        dref = {kk: (None if v is None else v.to(DEV)) for kk, v in bt.items()}
        ref, rnamed, _ = _make_step(cfg, gs, dref, True, False, 0.2)
        ref.seed_cell.fill_(500 + k)
        ref.run()
        rl, rg, rgr = _snapshot(ref, rnamed, live)
        ref.close()
        l, g, gr = got[k]
        assert np.isfinite(l) and l == rl, (seq[k], l, rl)
        assert np.array_equal(g, rg), seq[k]
        for n in live:
            assert np.array_equal(gr[n], rgr[n]), (seq[k], n, _rel(gr[n], rgr[n]))


@pytest.mark.parametrize("cfg_name,B,kind,seed", [("P19", 256, "ones", 100), ("P19", 256, "sparse", 7), ("P19", 37, "sparse", 3),
                                                  ("P12", 24, "sparse", 5)])
def test_benchmarked_step_against_float64(cfg_name, B, kind, seed):
    """The path `bench.py` times -- ONE hipGraph on the token plan, fused K1 / attention / chains / head, split-bf16 contractions --
    against the restatement of code/models_rd.py:278-387 + code/Raindrop.py:319-323 evaluated in FLOAT64 on the same fp32
    parameters and inputs (B = 256, seed 100, Setting-1 is the benchmark batch itself).  Every kernel that exists in the bf16 modes
    only (the shape-specialised K1, the fused chains, the plan) is thereby tied to exact arithmetic, not to another split-bf16
    kernel.  Bounds: logits 2e-5 abs (north star: 1e-4), loss 2e-6, every gradient 1e-3 in relative L2 -- the split products
    carry ~2^-16 per term and a ReLU gate whose pre-activation lies within that of zero may open on one side only (tests/
    test_gpu_parity.py `_grad_close`); measured values are printed (run with -s).
    P12 (round 5): the plan step BEYOND the P19 envelope -- T = 215: multi-tile attention on plan rows, the unfused message passing's
    panel products with the plan-following scatter, the streamed weight gradients -- had only been tied to its own padded form in the
    same arithmetic and to the goldens at 2 %; same bounds here."""
    from oracle import restatement as O2
    cfg = synth.make_config(cfg_name)
    gs = synth.make_structure(cfg, kind)
    batch = synth.make_batch(cfg, B, seed=seed)
    losses, logits, grads, step = _run_step(cfg, gs, batch, True, True)
    step.close()
    live = synth.live_parameter_names(cfg)
    m = build_ours(cfg, gs, "cpu", 7)
    p64 = {n: t.detach().double().requires_grad_(True) for n, t in m.named_parameters() if n in set(live)}
    b64 = {k: (v.double() if (v is not None and v.is_floating_point()) else v) for k, v in batch.items()}
    lg, ls, gr = O2.step_fwd_bwd(p64, cfg, b64, gs.double(), faithful=False)
    elog = float(np.abs(logits.astype(np.float64) - lg.numpy()).max())
    eloss = abs(losses[0] - float(ls))
    worst = ("", 0.0)
    for n in live:
        ref = gr[n].numpy()
        e = float(np.linalg.norm((grads[n].astype(np.float64) - ref).ravel()) / (np.linalg.norm(ref.ravel()) + 1e-300))
        if e > worst[1]:
            worst = (n, e)
    print("float64 tie %s B=%d %s: logits %.3e, loss %.3e, worst gradient rel-L2 %.3e (%s)" % (cfg_name, B, kind, elog, eloss, worst[1], worst[0]))
    assert elog < 2e-5 and eloss < 2e-6, (elog, eloss)
    assert worst[1] < 1e-3, worst
