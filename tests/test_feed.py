"""Batch feed: host index plan vs the oracle / the reference's own lines (CPU), and the
device gather vs numpy fancy indexing, bit for bit (GPU)."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import feed as O
from raindrop_amd import feed, synth

REF = "/root/reference/code/Raindrop.py"


@pytest.mark.parametrize("strategy,bs", [(2, 128), (2, 6), (3, 128), (1, 8)])
def test_index_plan_matches_oracle(strategy, bs):
    rng = np.random.default_rng(3)
    y = (rng.random(1000) < 0.1).astype(np.int64)
    np.random.seed(11)
    got = feed.epoch_index_plan(y, bs, strategy)
    np.random.seed(11)
    exp = O.epoch_batches(y, bs, strategy)
    assert len(got) == len(exp) > 0
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)
    if strategy == 2:                       # balanced halves, negatives first (Raindrop.py:303-305)
        assert all((y[b[:bs // 2]] == 0).all() and (y[b[bs // 2:]] == 1).all() for b in got)
        assert len(got) == min((y == 0).sum() // (bs // 2), 3 * (y == 1).sum() // (bs // 2))


@pytest.mark.skipif(not os.path.isfile(REF), reason="reference tree not present (GPU box)")
def test_index_plan_matches_reference_script_lines():
    """Pins the oracle: executes the reference's OWN statements for the strategy-2 plan (the index
    bookkeeping of `code/Raindrop.py:261-307`, extracted at test time, never copied into the repo)."""
    src = open(REF).read().splitlines()
    want = [r"^\s*idx_0 = np\.where\(ytrain == 0\)\[0\]", r"^\s*idx_1 = np\.where\(ytrain == 1\)\[0\]",
            r"^\s*n0, n1 = len\(idx_0\), len\(idx_1\)", r"^\s*expanded_idx_1 = np\.concatenate",
            r"^\s*expanded_n1 = len\(expanded_idx_1\)", r"^\s*K0 = n0 //", r"^\s*K1 = expanded_n1 //",
            r"^\s*n_batches = np\.min\(\[K0, K1\]\)", r"^\s*np\.random\.shuffle\(expanded_idx_1\)", r"^\s*I1 = expanded_idx_1",
            r"^\s*np\.random\.shuffle\(idx_0\)", r"^\s*I0 = idx_0"]
    body = []
    for pat in want:
        hits = [ln.strip() for ln in src if re.match(pat, ln)]
        assert hits, pat
        body.append(hits[0])
    per_batch = [ln.strip() for ln in src if re.match(r"^\s*idx[01]_batch = I[01]\[n \* int\(batch_size / 2\)", ln)]
    cat = [ln.strip() for ln in src if re.match(r"^\s*idx = np\.concatenate\(\[idx0_batch, idx1_batch\], axis=0\)", ln)]
    assert len(per_batch) == 2 and cat
    rng = np.random.default_rng(5)
    ytrain = (rng.random(777) < 0.07).astype(np.int64)
    env = {"np": np, "ytrain": ytrain, "batch_size": 128}
    np.random.seed(23)
    exec("\n".join(body), env)
    ref_batches = []
    for n in range(int(env["n_batches"])):
        env["n"] = n
        exec("\n".join(per_batch + [cat[0]]), env)
        ref_batches.append(env["idx"].copy())
    np.random.seed(23)
    mine = O.epoch_batches(ytrain, 128, 2)
    assert len(mine) == len(ref_batches) > 0
    for a, b in zip(mine, ref_batches):
        assert np.array_equal(a, b)


@pytest.mark.skipif(not os.path.isfile(REF), reason="reference tree not present (GPU box)")
def test_epoch_planner_matches_reference_script_over_several_epochs():
    """The script shuffles `expanded_idx_1` / `idx_0` IN PLACE every epoch (`code/Raindrop.py:292-296`): epoch k's order is
    a permutation of epoch k-1's.  `feed.EpochPlanner` must reproduce the batches of every epoch, not only the first;
    checked against the reference's own statements, extracted at test time and executed in the script's nesting."""
    src = open(REF).read().splitlines()
    def grab(pat):
        hits = [ln.strip() for ln in src if re.match(pat, ln)]
        assert hits, pat
        return hits[0]
    setup = [grab(p) for p in (r"^\s*idx_0 = np\.where\(ytrain == 0\)\[0\]", r"^\s*idx_1 = np\.where\(ytrain == 1\)\[0\]",
                               r"^\s*n0, n1 = len\(idx_0\), len\(idx_1\)", r"^\s*expanded_idx_1 = np\.concatenate",
                               r"^\s*expanded_n1 = len\(expanded_idx_1\)", r"^\s*K0 = n0 //", r"^\s*K1 = expanded_n1 //",
                               r"^\s*n_batches = np\.min\(\[K0, K1\]\)")]
    per_epoch = [grab(p) for p in (r"^\s*np\.random\.shuffle\(expanded_idx_1\)", r"^\s*I1 = expanded_idx_1",
                                   r"^\s*np\.random\.shuffle\(idx_0\)", r"^\s*I0 = idx_0")]
    per_batch = [ln.strip() for ln in src if re.match(r"^\s*idx[01]_batch = I[01]\[n \* int\(batch_size / 2\)", ln)]
    per_batch.append(grab(r"^\s*idx = np\.concatenate\(\[idx0_batch, idx1_batch\], axis=0\)"))
    rng = np.random.default_rng(8)
    ytrain = (rng.random(900) < 0.08).astype(np.int64)
    env = {"np": np, "ytrain": ytrain, "batch_size": 64}
    np.random.seed(31)
    exec("\n".join(setup), env)
    ref = []
    for epoch in range(3):
        exec("\n".join(per_epoch), env)
        for n in range(int(env["n_batches"])):
            env["n"] = n
            exec("\n".join(per_batch), env)
            ref.append(env["idx"].copy())
    np.random.seed(31)
    planner = feed.EpochPlanner(ytrain, batch_size=64, strategy=2)
    mine = [b for _ in range(3) for b in planner.next_epoch()]
    assert len(mine) == len(ref) == 3 * planner.n_batches > 0
    for a, b in zip(mine, ref):
        assert np.array_equal(a, b)
    np.random.seed(31)
    first = feed.epoch_index_plan(ytrain, 64, 2)                 # the one-epoch helper: first epoch only
    assert all(np.array_equal(a, b) for a, b in zip(first, ref[: len(first)]))


def test_gather_oracle_is_fancy_indexing():
    rng = np.random.default_rng(0)
    P = rng.standard_normal((5, 9, 4)).astype(np.float32)
    Tm = np.maximum(rng.standard_normal((5, 9)), 0).astype(np.float32)
    idx = np.array([8, 0, 0, 3])
    Pb, Sb, Tb, yb, ln = O.gather_batch(P, Tm, None, None, idx)
    assert Pb.shape == (5, 4, 4) and Sb is None and yb is None
    assert np.array_equal(ln, (Tm[:, idx] > 0).sum(0)) and ln.dtype == np.int64


def test_device_dataset_refuses_cpu():
    from raindrop_amd import _lib
    with pytest.raises(_lib.RaindropHipError):
        feed.DeviceDataset(np.zeros((2, 3, 4), np.float32), np.zeros((2, 3), np.float32), device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("cfg_name,N,B", [("P19", 700, 256), ("PAM", 40, 17), ("TINY", 9, 1), ("P12", 64, 128)])
def test_batch_gather_bit_exact(cfg_name, N, B):
    cfg = synth.make_config(cfg_name)
    b = synth.make_batch(cfg, N, seed=5)
    P, Tm = b["src"].numpy(), b["times"].numpy()
    S = None if b["static"] is None else b["static"].numpy()
    y = b["y"].numpy()
    ds = feed.DeviceDataset(P, Tm, S, y)
    rng = np.random.default_rng(B)
    idx = rng.integers(0, N, size=B)                    # with repeats, like the 3x up-sampled positives
    Pb, Sb, Tb, yb, ln = ds.batch(idx)
    eP, eS, eT, ey, el = O.gather_batch(P, Tm, S, y, idx)
    assert ds.bad_indices() == 0
    assert np.array_equal(Pb.cpu().numpy(), eP) and np.array_equal(Tb.cpu().numpy(), eT)
    assert np.array_equal(ln.cpu().numpy(), el) and ln.dtype == torch.int64
    assert np.array_equal(yb.cpu().numpy(), ey)
    assert (Sb is None and eS is None) or np.array_equal(Sb.cpu().numpy(), eS)
    # buffers of a previous call are reused in place; device-side indices are clamped and counted
    out = ds.alloc(B)
    ds.batch(idx, out=out)
    assert np.array_equal(out["P"].cpu().numpy(), eP)
    bad = torch.tensor([N + 5] + [0] * (B - 1), dtype=torch.int64, device="cuda")
    ds.batch(bad, out=out)
    assert ds.bad_indices() == 1
    assert np.array_equal(out["P"][:, 0].cpu().numpy(), P[:, N - 1])
    with pytest.raises(IndexError):
        ds.batch(np.array([N]))


@pytest.mark.gpu
def test_feed_drives_the_model_and_chunked_eval_matches_one_forward():
    from tests.helpers import build_ours
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "ones")
    N = 300
    b = synth.make_batch(cfg, N, seed=8)
    ds = feed.DeviceDataset(b["src"], b["times"], b["static"], b["y"])
    m = build_ours(cfg, gs, "cuda", 3).eval()
    np.random.seed(1)
    plan = feed.epoch_index_plan(b["y"].numpy(), batch_size=32, strategy=2)
    P, S, Tm, y, ln = ds.batch(plan[0])
    with torch.no_grad():
        out, _, _ = m(P, S, Tm, ln)
    assert out.shape == (32, 2) and torch.isfinite(out).all()
    full = feed.evaluate_chunked(m, ds, chunk=N)        # utils_rd.evaluate_standard: one forward
    parts = feed.evaluate_chunked(m, ds, chunk=128)
    assert full.shape == (N, 2) and torch.equal(full, parts)
