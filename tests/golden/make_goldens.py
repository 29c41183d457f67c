"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF (oracle O1).

Run in the build container only (needs `/root/reference`):  python tests/golden/make_goldens.py

The reference's own files (`code/models_rd.py`, `code/Ob_propagation.py`, `code/transformer_conv.py`)
are executed unmodified on CPU under `oracle/ref_loader.py`.  Inputs and weights are NOT stored:
they are regenerated bit-identically from seeds by `raindrop_amd.synth` (numpy PCG64), so the
fixtures stay small.  Stored per model case:
  logits (train mode with every dropout p forced to 0, and eval mode), CE loss, `distance`,
  the INT artefacts (padding mask, edge_index, edge_weights), strided samples of the
  message-passing output / PE, and for every live parameter gradient: sum, L2 norm and a strided
  sample (full tensor when it has <= 70k elements).
Each case also records how well the independent restatement (oracle O2) agreed with O1 when the
fixture was made; the generator refuses to write a fixture if they disagree.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_loader, restatement as O2      # noqa: E402
from raindrop_amd import synth                        # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLE = 4096

MODEL_CASES = [
    # (fixture name, config, batch, structure kind, param seed, batch seed)
    ("tiny_sparse", "TINY", 3, "sparse", 1, 3),
    ("p19_ones", "P19", 32, "ones", 2, 4),
    ("p19_sparse", "P19", 32, "sparse", 3, 5),
    ("p12_ones", "P12", 8, "ones", 4, 6),
    ("pam_ones", "PAM", 4, "ones", 5, 7),
    # the benchmark's batch (B = 256: the B*F = 8704-row weight-gradient reduction, every dW slice of the fused path) and a
    # larger P12 batch on the generic-K path; gradients stored as strided samples only (FULL_LIMIT) to stay small
    ("p19_b256", "P19", 256, "ones", 6, 8),
    ("p12_b32", "P12", 32, "sparse", 7, 9),
    # BASELINE.json configs[4], the 256-sensor x 512-step stress shape (K = 2048 message-passing features, two attention
    # heads of 520): two samples are what the CPU reference finishes in minutes
    ("syn256_b2", "SYN256", 2, "sparse", 8, 10),
]
FULL_LIMIT = {"p19_x3_sparse": 4096, "p19_b256": 4096, "p12_b32": 4096, "syn256_b2": 4096, "p19_beta_sparse": 4096, "p19_beta_ones": 4096,
              "p12_beta_sparse": 4096, "wide80_beta_sparse": 4096}
# the paper's branch: the reference with its `use_beta = False` literal (code/models_rd.py:317) flipped IN MEMORY by
# oracle/ref_loader.load_models_rd_with_beta(); Raindrop_v2(use_beta=True, compute_distance=True) here
BETA_CASES = [
    ("p19_beta_sparse", "P19", 8, "sparse", 11, 12),
    ("p19_beta_ones", "P19", 8, "ones", 13, 14),
    ("p12_beta_sparse", "P12", 3, "sparse", 15, 16),
    ("wide80_beta_sparse", "WIDE80", 3, "sparse", 17, 18),     # 80 sensors: the graph operator's workspace form at model level
]


# Weights at 3x the init scale (matrices ~ U(+-3/sqrt(fan_in))): activations, scores and gradients an order of magnitude above
# the other fixtures' -- where the margin of the split-bf16 products and of the softmax / LayerNorm rounding would show first
# (reduced-precision margins "must be re-checked" away from init-scale weights).
SCALED_CASES = [("p19_x3_sparse", "P19", 32, "sparse", 31, 32, 3.0)]
# Trajectory: TRAJ_STEPS steps of the reference's own loop body (code/Raindrop.py:319-324: forward, zero_grad, CrossEntropyLoss,
# backward, Adam step) on TRAJ_STEPS different batches, dropout 0; lr 1e-3 (10x the script's) so that 20 steps MOVE the weights.
TRAJ_STEPS, TRAJ_LR, TRAJ_B = 20, 1e-3, 32


def traj_case(name="p19_traj20", cfg_name="P19", kind="sparse", pseed=41, bseed0=300):
    """Trained-weight parity (VERDICT r5 missing #3): the state after TRAJ_STEPS optimizer steps of the REFERENCE's model under the
    reference's loop body, every step on a new batch.  Stored: the loss of every step, the logits of the last step's forward, the
    final value of every live parameter (in full), and -- on a held-out batch under those weights -- the eval- and train-mode
    logits, the loss and every gradient (strided samples)."""
    cfg = synth.make_config(cfg_name)
    gs = synth.make_structure(cfg, kind)
    model = ref_loader.build_raindrop_v2(cfg, gs.clone())
    synth.fill_params_(model, seed=pseed)
    zero_dropout(model)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=TRAJ_LR)                  # code/Raindrop.py:256
    crit = torch.nn.CrossEntropyLoss()                                     # :255
    losses, last = [], None
    for i in range(TRAJ_STEPS):
        b = synth.make_batch(cfg, TRAJ_B, seed=bseed0 + i)
        outputs, _, _ = ref_loader.forward(model, b["src"], b["static"], b["times"], b["lengths"])   # :319
        opt.zero_grad()                                                    # :321
        loss = crit(outputs, b["y"])                                       # :322
        loss.backward()                                                    # :323
        opt.step()                                                         # :324
        losses.append(float(loss))
        last = outputs.detach()
    held = synth.make_batch(cfg, TRAJ_B, seed=bseed0 + 1000)
    model.eval()
    with torch.no_grad():
        held_logits, _, _ = ref_loader.forward(model, held["src"], held["static"], held["times"], held["lengths"])
    # one more forward + backward at the TRAINED weights (no optimizer step): the gradients a 21st step would see
    model.train()
    opt.zero_grad()
    hl_train, _, _ = ref_loader.forward(model, held["src"], held["static"], held["times"], held["lengths"])
    held_loss = crit(hl_train, held["y"])
    held_loss.backward()
    params = dict(model.named_parameters())
    live = synth.live_parameter_names(cfg)
    init = {n: synth.param_values(n, params[n].shape, pseed) for n in live}
    out = dict(meta=json.dumps(dict(name=name, cfg=cfg_name, batch=TRAJ_B, structure=kind, param_seed=pseed, batch_seed0=bseed0,
                                    held_out_seed=bseed0 + 1000, steps=TRAJ_STEPS, lr=TRAJ_LR, torch=torch.__version__)),
               losses=np.array(losses, dtype=np.float64), last_logits=last.numpy(), held_logits=held_logits.numpy(),
               held_logits_train=hl_train.detach().numpy(), held_loss=np.float32(held_loss.item()), live=np.array(live))
    for n in live:
        # the FULL trained tensor (0.5 M floats in all): a free-running replay in a reduced-precision mode drifts (Adam turns a
        # 1e-3 relative gradient error on a small entry into an O(lr) difference of its update), so the tests also load these
        # weights and compare logits / gradients AT them
        out["trained/" + n] = params[n].detach().numpy().copy()
        out["wmoved/" + n] = np.float64((params[n].detach() - init[n]).double().norm().item())   # how far 20 steps moved it
        g_ = params[n].grad
        s_, st = strided(g_, full_limit=4096)
        out["heldgrad/" + n] = s_
        out["heldgradstride/" + n] = np.int64(st)
        out["heldgradnorm/" + n] = np.float64(g_.double().norm().item())
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-12s %d steps, loss %.5f -> %.5f, moved %s -> %s (%.1f KB)" % (
        name, TRAJ_STEPS, losses[0], losses[-1], {n.split(".")[-2] if "." in n else n: "%.3g" % out["wmoved/" + n] for n in live[:4]},
        os.path.basename(path), os.path.getsize(path) / 1024))


def strided(t, n=SAMPLE, full_limit=70_000):
    flat = t.detach().reshape(-1)
    if flat.numel() <= full_limit:
        return flat.numpy().copy(), 1
    stride = max(1, flat.numel() // n)
    return flat[::stride].numpy().copy(), stride


def zero_dropout(model):
    """Parity runs use dropout = identity (train-mode RNG streams cannot match across devices)."""
    for mod in model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0


def model_case(name, cfg_name, B, kind, pseed, bseed, use_beta=False, scale=1.0):
    cfg = synth.make_config(cfg_name)
    gs = synth.make_structure(cfg, kind)
    model = ref_loader.build_raindrop_v2(cfg, gs.clone(), use_beta=use_beta)
    synth.fill_params_(model, seed=pseed, scale=scale)
    zero_dropout(model)
    b = synth.make_batch(cfg, B, seed=bseed)

    model.train()
    logits, distance, _ = ref_loader.forward(model, b["src"], b["static"], b["times"], b["lengths"])
    loss = F.cross_entropy(logits, b["y"])
    loss.backward()
    model.eval()
    with torch.no_grad():
        logits_eval, _, _ = ref_loader.forward(model, b["src"], b["static"], b["times"], b["lengths"])

    params = dict(model.named_parameters())
    live = [n for n, t in params.items() if t.grad is not None]
    want_live = synth.live_parameter_names(cfg) + (["ob_propagation.map_weights", "ob_propagation.increase_dim.weight",
                                                   "ob_propagation.increase_dim.bias"] if use_beta else [])
    assert sorted(live) == sorted(want_live), (live, want_live)

    # cross-check with the independent restatement before trusting either
    p = {n: t.detach().clone().requires_grad_(n in live) for n, t in params.items()}
    lg2, loss2, g2 = O2.step_fwd_bwd(p, cfg, b, gs, faithful=use_beta, use_beta=use_beta)
    _, d2, inter = O2.raindrop_v2_forward({n: t.detach() for n, t in params.items()}, cfg, b["src"],
                                          b["static"], b["times"], b["lengths"], gs, faithful=use_beta,
                                          return_intermediates=True, use_beta=use_beta)
    assert abs(float(d2) - float(distance)) <= 1e-6 * max(1.0, abs(float(distance))), (float(d2), float(distance))
    e_logit = float((lg2 - logits.detach()).abs().max())
    e_grad = max(float((g2[n] - params[n].grad).abs().max() / (params[n].grad.abs().max() + 1e-30))
                 for n in live)
    # use_beta: the pruned edges of a sample are summed in pruning order, and the reference's argsort (unstable) orders tied
    # scores (all-ones structure: 34 equal edges per target) differently from the restatement's -- rounding-level, bounded looser
    assert e_logit < 2e-6 and e_grad < (1e-4 if use_beta else 2e-5), (name, e_logit, e_grad)

    ei, ew = O2.build_graph(gs.numpy())
    out = dict(
        meta=json.dumps(dict(name=name, cfg=cfg_name, batch=B, structure=kind, param_seed=pseed, param_scale=scale,
                             batch_seed=bseed, o2_vs_o1_logit=e_logit, o2_vs_o1_grad_rel=e_grad,
                             torch=torch.__version__)),
        logits=logits.detach().numpy(), logits_eval=logits_eval.numpy(),
        loss=np.float32(loss.item()), distance=np.float32(float(distance)),
        mask=O2.padding_mask(b["lengths"].numpy(), cfg["max_len"]),
        edge_index=ei, edge_weights=ew, lengths=b["lengths"].numpy(),
        live=np.array(live),
    )
    fl = FULL_LIMIT.get(name, 70_000)
    for key in ("msg_out", "pe", "agg"):
        s, st = strided(inter[key], full_limit=fl)
        out["inter_" + key] = s
        out["inter_" + key + "_stride"] = np.int64(st)
    for n in live:
        g = params[n].grad
        s, st = strided(g, full_limit=fl)
        out["grad/" + n] = s
        out["gradstride/" + n] = np.int64(st)
        out["gradsum/" + n] = np.float64(g.double().sum().item())
        out["gradnorm/" + n] = np.float64(g.double().norm().item())
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-12s logits %s loss %.6f  O2-vs-O1 logit %.2e grad %.2e  -> %s (%.1f KB)" % (
        name, tuple(logits.shape), loss.item(), e_logit, e_grad, os.path.basename(path),
        os.path.getsize(path) / 1024))


def operator_cases():
    """Stand-alone operator fixtures: `Observation_progation` (default and use_beta branches) and
    `TransformerConv`, run from the reference classes on small graphs; full tensors are stored
    (weights included -- they are tiny)."""
    ref = ref_loader.load()
    rng = np.random.default_rng(123)
    out = {}
    # --- Observation_progation, default branch: N=6 nodes, T=5 steps, ob_dim=4 -> K=20
    n, T, d = 6, 5, 4
    K = T * d
    op = ref.run(ref.Ob_propagation.Observation_progation, in_channels=K, out_channels=K, heads=1,
                 n_nodes=n, ob_dim=d)
    synth.fill_params_(op, seed=11)
    adj = (rng.random((n, n)) * (rng.random((n, n)) < 0.5)).astype(np.float32)
    ei, ew = O2.build_graph(adj)
    x = torch.from_numpy(rng.standard_normal((n, K)).astype(np.float32))
    p_t = torch.from_numpy(rng.standard_normal((T, 16)).astype(np.float32))
    y, (ei_o, alpha) = ref.run(op.forward, x, p_t=p_t, edge_index=torch.from_numpy(ei),
                               edge_weights=torch.from_numpy(ew), use_beta=False, edge_attr=None,
                               return_attention_weights=True)
    out.update(obp_adj=adj, obp_x=x.numpy(), obp_p_t=p_t.numpy(), obp_y=y.detach().numpy(),
               obp_alpha=alpha.detach().numpy(), obp_ei=ei_o.numpy(),
               obp_w=op.lin_value.weight.detach().numpy(), obp_b=op.lin_value.bias.detach().numpy())
    # --- use_beta branch (dead by flag in the model; general prune/normalise-by-source operator)
    yb, (ei_b, alpha_b) = ref.run(op.forward, x, p_t=p_t, edge_index=torch.from_numpy(ei),
                                  edge_weights=torch.from_numpy(ew), use_beta=True, edge_attr=None,
                                  return_attention_weights=True)
    out.update(obpb_y=yb.detach().numpy(), obpb_alpha=alpha_b.detach().numpy(), obpb_ei=ei_b.numpy(),
               obpb_w_inc=op.increase_dim.weight.detach().numpy(),
               obpb_b_inc=op.increase_dim.bias.detach().numpy(),
               obpb_map=op.map_weights.detach().numpy())
    # --- TransformerConv with edge weights: N=7 nodes, 9 -> 12 channels
    n2, cin, cout = 7, 9, 12
    tc = ref.run(ref.transformer_conv.TransformerConv, in_channels=cin, out_channels=cout, heads=1)
    synth.fill_params_(tc, seed=12)
    adj2 = (rng.random((n2, n2)) * (rng.random((n2, n2)) < 0.6)).astype(np.float32)
    ei2, ew2 = O2.build_graph(adj2)
    x2 = torch.from_numpy(rng.standard_normal((n2, cin)).astype(np.float32))
    y2, (_, alpha2) = ref.run(tc.forward, x2, edge_index=torch.from_numpy(ei2),
                              edge_weights=torch.from_numpy(ew2), edge_attr=None,
                              return_attention_weights=True)
    out.update(tc_adj=adj2, tc_x=x2.numpy(), tc_y=y2.detach().numpy(), tc_alpha=alpha2.detach().numpy(),
               tc_wv=tc.lin_value.weight.detach().numpy(), tc_bv=tc.lin_value.bias.detach().numpy(),
               tc_ws=tc.lin_skip.weight.detach().numpy(), tc_bs=tc.lin_skip.bias.detach().numpy())
    # O2 must reproduce all three before the fixture is written
    t = torch.from_numpy
    y_o2, a_o2 = O2.observation_propagation(x, t(ei), t(ew), op.lin_value.weight.detach(),
                                            op.lin_value.bias.detach())
    assert float((y_o2 - y.detach()).abs().max()) < 1e-6 and torch.equal(a_o2, alpha.detach())
    yb_o2, (eib_o2, ab_o2) = O2.observation_propagation_beta(
        x, p_t, t(ei), t(ew), op.lin_value.weight.detach(), op.lin_value.bias.detach(),
        op.increase_dim.weight.detach(), op.increase_dim.bias.detach(), op.map_weights.detach(), d)
    assert float((yb_o2 - yb.detach()).abs().max()) < 1e-6 and torch.equal(eib_o2, ei_b)
    y2_o2, a2_o2 = O2.transformer_conv(x2, t(ei2), t(ew2), tc.lin_value.weight.detach(),
                                       tc.lin_value.bias.detach(), tc.lin_skip.weight.detach(),
                                       tc.lin_skip.bias.detach())
    assert float((y2_o2 - y2.detach()).abs().max()) < 1e-6
    assert float((a2_o2 - alpha2.detach()).abs().max()) < 1e-7
    path = os.path.join(HERE, "operators.npz")
    np.savez_compressed(path, **out)
    print("operators    -> %s (%.1f KB)" % (os.path.basename(path), os.path.getsize(path) / 1024))


TCONV_GENERAL = [   # (tag, heads, channels, concat, edge_dim, beta, root_weight, param seed)
    ("a", 2, 6, True, None, False, True, 51),
    ("b", 3, 4, False, 5, True, True, 52),
    ("c", 1, 8, True, 3, False, False, 53),
]


def tconv_general_case():
    """The reference's `TransformerConv` WITHOUT given edge weights (code/transformer_conv.py:139-207: q.k scores, heads, lin_edge on
    the key, concat / mean, root weight with and without the beta gate) on a 9-node graph with a duplicated edge, eval-free
    (dropout 0): output, post-softmax alpha and the gradients of <y, R> w.r.t. x, edge_attr and every parameter.  O2's
    `transformer_conv_general` must reproduce them before the fixture is written."""
    ref = ref_loader.load()
    rng = np.random.default_rng(321)
    n, cin = 9, 7
    adj = (rng.random((n, n)) < 0.45)
    src, tgt = np.nonzero(adj)
    ei = np.stack([np.concatenate([src, src[:2]]), np.concatenate([tgt, tgt[:2]])]).astype(np.int64)      # two duplicated edges
    out = dict(ei=ei, x=rng.standard_normal((n, cin)).astype(np.float32))
    for tag, H, C, concat, edim, beta, root, pseed in TCONV_GENERAL:
        tc = ref.run(ref.transformer_conv.TransformerConv, in_channels=cin, out_channels=C, heads=H, concat=concat, beta=beta,
                     edge_dim=edim, root_weight=root)
        synth.fill_params_(tc, seed=pseed)
        x = torch.from_numpy(out["x"]).requires_grad_(True)
        ea = None if edim is None else torch.from_numpy(rng.standard_normal((ei.shape[1], edim)).astype(np.float32)).requires_grad_(True)
        y, (_, alpha) = ref.run(tc.forward, x, edge_index=torch.from_numpy(ei), edge_weights=None, edge_attr=ea,
                                return_attention_weights=True)
        R = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
        names = [k for k, _ in tc.named_parameters()]
        wrt = [x] + ([ea] if ea is not None else []) + [p_ for _, p_ in tc.named_parameters()]
        grads = torch.autograd.grad((y * R).sum(), wrt, allow_unused=True)
        # O2 before trusting either
        p2 = {k: v.detach() for k, v in tc.named_parameters()}
        y2, a2 = O2.transformer_conv_general(x.detach(), torch.from_numpy(ei), p2, H, C, concat, None if ea is None else ea.detach(), root)
        assert float((y2 - y.detach()).abs().max()) < 2e-6 and float((a2 - alpha.detach()).abs().max()) < 1e-6, tag
        out.update({tag + "_y": y.detach().numpy(), tag + "_alpha": alpha.detach().numpy(), tag + "_R": R.numpy(),
                    tag + "_gx": grads[0].numpy()})
        gi = 1
        if ea is not None:
            out[tag + "_ea"] = ea.detach().numpy(); out[tag + "_gea"] = grads[1].numpy(); gi = 2
        for k, g_ in zip(names, grads[gi:]):
            if g_ is not None:
                out[tag + "_g/" + k] = g_.numpy()
    path = os.path.join(HERE, "tconv_general.npz")
    np.savez_compressed(path, **out)
    print("tconv_general -> %s (%.1f KB)" % (os.path.basename(path), os.path.getsize(path) / 1024))


def beta_batched_case():
    """The use_beta operator applied to B sample graphs that share an edge list (what `Raindrop_v2.forward` would do per
    sample with `use_beta=True`, code/models_rd.py:313-343), by the REFERENCE class: outputs, pruned edge lists, returned
    scores, gradients of <y, R> w.r.t. the inputs and the four live parameter tensors, and the structure distance of the
    returned scores (code/models_rd.py:345-346)."""
    ref = ref_loader.load()
    rng = np.random.default_rng(321)
    n, T, d, B = 12, 15, 4, 5
    K = T * d
    op = ref.run(ref.Ob_propagation.Observation_progation, in_channels=K, out_channels=K, heads=1, n_nodes=n, ob_dim=d)
    synth.fill_params_(op, seed=21)
    adj = (rng.random((n, n)) * (rng.random((n, n)) < 0.6)).astype(np.float32)
    ei, ew = O2.build_graph(adj)
    X = torch.from_numpy(rng.standard_normal((B, n, K)).astype(np.float32)).requires_grad_(True)
    PT = torch.from_numpy(rng.standard_normal((B, T, 16)).astype(np.float32))
    R = torch.from_numpy(rng.standard_normal((B, n, K)).astype(np.float32))
    ys, eis, alphas = [], [], []
    for b in range(B):
        y, (ei_b, a_b) = ref.run(op.forward, X[b], p_t=PT[b], edge_index=torch.from_numpy(ei), edge_weights=torch.from_numpy(ew),
                                 use_beta=True, edge_attr=None, return_attention_weights=True)
        ys.append(y); eis.append(ei_b); alphas.append(a_b)
    Y = torch.stack(ys)
    params = [op.lin_value.weight, op.lin_value.bias, op.increase_dim.weight, op.increase_dim.bias, op.map_weights]
    grads = torch.autograd.grad((Y * R).sum(), [X] + params)
    alpha_all = torch.stack([a.detach() for a in alphas], dim=1)                    # [Kk, B]
    dist = torch.mean(torch.cdist(alpha_all.T, alpha_all.T, p=2))
    # the restatement must agree sample by sample
    for b in range(B):
        y2, (ei2, a2) = O2.observation_propagation_beta(X[b].detach(), PT[b], torch.from_numpy(ei), torch.from_numpy(ew),
                                                        op.lin_value.weight.detach(), op.lin_value.bias.detach(),
                                                        op.increase_dim.weight.detach(), op.increase_dim.bias.detach(),
                                                        op.map_weights.detach(), d)
        assert float((y2 - ys[b].detach()).abs().max()) < 1e-6 and torch.equal(ei2, eis[b])
    out = dict(adj=adj, X=X.detach().numpy(), PT=PT.numpy(), R=R.numpy(), Y=Y.detach().numpy(),
               ei=torch.stack(eis).numpy(), alpha=torch.stack([a.detach() for a in alphas]).numpy(), distance=np.float32(dist.item()),
               gX=grads[0].numpy(), gWv=grads[1].numpy(), gbv=grads[2].numpy(), gWi=grads[3].numpy(), gbi=grads[4].numpy(),
               gmap=grads[5].numpy(), dims=np.array([n, T, d, B]))
    path = os.path.join(HERE, "beta_batched.npz")
    np.savez_compressed(path, **out)
    print("beta_batched -> %s (%.1f KB), distance %.6f" % (os.path.basename(path), os.path.getsize(path) / 1024, dist.item()))


def beta_large_case():
    """The use_beta operator on a graph that does not fit one workgroup's LDS -- 256 nodes (SYN256's sensor count), 13 360 edges
    (sparse random structure with distinct weights: no pruning ties), T = 6 steps -- by the REFERENCE class, per sample:
    outputs, pruned edge lists, returned scores, gradients of <y, R>.  Exercises the workspace form of rd_graph_beta_fwd / _bwd
    (rd_graph_beta_large.hip): a 16 384-key sort in 4096-key chunks, per-node lists over 6 680 kept edges."""
    ref = ref_loader.load()
    n, T, d, B = 256, 6, 4, 2
    K = T * d
    op = ref.run(ref.Ob_propagation.Observation_progation, in_channels=K, out_channels=K, heads=1, n_nodes=n, ob_dim=d)
    synth.fill_params_(op, seed=31)
    # ~13 k float32 scores collide now and then, and the reference's argsort (unstable) orders tied edges arbitrarily: take the
    # first seed of a fixed list whose scores are all distinct, so that the pruned edge LIST is reproducible bit for bit
    for seed in range(654, 700):
        rng = np.random.default_rng(seed)
        adj = (rng.uniform(0.5, 1.5, (n, n)) * (rng.random((n, n)) < 0.2)).astype(np.float32)
        ei, ew = O2.build_graph(adj)
        X = torch.from_numpy((rng.standard_normal((B, n, K)) * 0.5).astype(np.float32)).requires_grad_(True)
        PT = torch.from_numpy(rng.standard_normal((B, T, 16)).astype(np.float32))
        R = torch.from_numpy(rng.standard_normal((B, n, K)).astype(np.float32))
        ys, eis, alphas = [], [], []
        for b in range(B):
            y, (ei_b, a_b) = ref.run(op.forward, X[b], p_t=PT[b], edge_index=torch.from_numpy(ei), edge_weights=torch.from_numpy(ew),
                                     use_beta=True, edge_attr=None, return_attention_weights=True)
            ys.append(y); eis.append(ei_b); alphas.append(a_b)
        # ties anywhere in the FULL score list matter (a tie across the pruning boundary changes the kept set): recompute them all
        full = [O2.beta_edge_scores(X[b].detach(), PT[b], torch.from_numpy(ei), torch.from_numpy(ew), op.increase_dim.weight.detach(),
                                    op.increase_dim.bias.detach(), op.map_weights.detach(), d).numpy() for b in range(B)]
        if all(np.unique(f).size == f.size for f in full):
            break
        print("  seed %d: tied scores, next" % seed)
    else:
        raise SystemExit("beta_large: no tie-free seed")
    Y = torch.stack(ys)
    params = [op.lin_value.weight, op.lin_value.bias, op.increase_dim.weight, op.increase_dim.bias, op.map_weights]
    grads = torch.autograd.grad((Y * R).sum(), [X] + params)
    for b in range(B):
        a = alphas[b].detach().numpy().ravel()
        assert np.all(np.diff(a) < 0), "tied scores: the reference's argsort order would not be reproducible"
        y2, (ei2, a2) = O2.observation_propagation_beta(X[b].detach(), PT[b], torch.from_numpy(ei), torch.from_numpy(ew),
                                                        op.lin_value.weight.detach(), op.lin_value.bias.detach(),
                                                        op.increase_dim.weight.detach(), op.increase_dim.bias.detach(),
                                                        op.map_weights.detach(), d)
        assert float((y2 - ys[b].detach()).abs().max()) < 1e-5 and torch.equal(ei2, eis[b])
    out = dict(adj=adj, X=X.detach().numpy(), PT=PT.numpy(), R=R.numpy(), Y=Y.detach().numpy(),
               ei=torch.stack(eis).numpy().astype(np.int32), alpha=torch.stack([a.detach() for a in alphas]).numpy(),
               gX=grads[0].numpy(), gWv=grads[1].numpy(), gbv=grads[2].numpy(), gWi=grads[3].numpy(), gbi=grads[4].numpy(),
               gmap=grads[5].numpy(), dims=np.array([n, T, d, B]))
    path = os.path.join(HERE, "beta_large.npz")
    np.savez_compressed(path, **out)
    print("beta_large   -> %s (%.1f KB), E = %d, kept %d" % (os.path.basename(path), os.path.getsize(path) / 1024, ei.shape[1],
                                                            eis[0].shape[1]))


def legacy_v1_case():
    """The legacy `Raindrop` class (code/models_rd.py:46-191; exported by `from models_rd import *`, never built by the script),
    run by the reference itself at the only shape its forward admits (215 steps, 36 sensors hard-coded at :150,:155):
    logits (train mode, dropout forced to 0, and eval), loss, distance, every live gradient, and the state_dict surface."""
    ref = ref_loader.load()
    d_inp, d_model, nhead, nhid, nlayers, T, d_static, B = 36, 72, 4, 96, 2, 215, 9, 3
    gs = synth.make_structure(dict(d_inp=d_inp), "sparse", seed=3)
    model = ref.run(ref.models_rd.Raindrop, d_inp, d_model, nhead, nhid, nlayers, 0.3, T, d_static, 100, 0.5, "mean", 2, gs.clone())
    synth.fill_params_(model, seed=31)
    zero_dropout(model)
    cfg = dict(max_len=T, d_inp=d_inp, static=True, d_static=d_static, n_classes=2)
    b = synth.make_batch(cfg, B, seed=32, density=0.4)
    model.train()
    logits, distance, _ = ref_loader.forward(model, b["src"], b["static"], b["times"], b["lengths"])
    loss = F.cross_entropy(logits, b["y"])
    loss.backward()
    model.eval()
    with torch.no_grad():
        logits_eval, _, _ = ref_loader.forward(model, b["src"], b["static"], b["times"], b["lengths"])
    params = dict(model.named_parameters())
    live = [n for n, t in params.items() if t.grad is not None]
    out = dict(meta=json.dumps(dict(d_inp=d_inp, d_model=d_model, nhead=nhead, nhid=nhid, nlayers=nlayers, max_len=T, d_static=d_static,
                                    batch=B, param_seed=31, batch_seed=32, density=0.4, structure_seed=3, torch=torch.__version__)),
               logits=logits.detach().numpy(), logits_eval=logits_eval.numpy(), loss=np.float32(loss.item()),
               distance=np.float32(float(distance)), live=np.array(live),
               surface=json.dumps({k: list(v.shape) for k, v in model.state_dict().items()}))
    for n in live:
        g = params[n].grad
        sm, st = strided(g, full_limit=4096)
        out["grad/" + n] = sm
        out["gradstride/" + n] = np.int64(st)
        out["gradnorm/" + n] = np.float64(g.double().norm().item())
    path = os.path.join(HERE, "legacy_v1.npz")
    np.savez_compressed(path, **out)
    print("legacy_v1    logits %s loss %.6f distance %.3g, %d live tensors -> %s (%.1f KB)" % (
        tuple(logits.shape), loss.item(), float(distance), len(live), os.path.basename(path), os.path.getsize(path) / 1024))


def state_dict_surface():
    """Names and shapes of the reference's state_dict per dataset config (checkpoint surface,
    code/Raindrop.py:374,381).  Under the CPU shim `R_u` is a registered parameter."""
    out = {}
    for cfg_name in ("TINY", "P19", "P12", "PAM", "SYN256", "WIDE80"):
        cfg = synth.make_config(cfg_name)
        model = ref_loader.build_raindrop_v2(cfg, synth.make_structure(cfg, "ones"))
        out[cfg_name] = {k: list(v.shape) for k, v in model.state_dict().items()}
        del model
    path = os.path.join(HERE, "state_dict_surface.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print("state_dict   -> %s" % os.path.basename(path))


if __name__ == "__main__":
    assert ref_loader.available(), "needs the reference tree (build container only)"
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    only = sys.argv[1:]
    if not only or "operators" in only:
        operator_cases()
    if not only or "state_dict" in only:
        state_dict_surface()
    if not only or "tconv_general" in only:
        tconv_general_case()
    if not only or "beta_batched" in only:
        beta_batched_case()
    if not only or "beta_large" in only:
        beta_large_case()
    if not only or "legacy_v1" in only:
        legacy_v1_case()
    for case in MODEL_CASES:
        if not only or case[0] in only:
            model_case(*case)
    for case in BETA_CASES:
        if not only or case[0] in only:
            model_case(*case, use_beta=True)
    for case in SCALED_CASES:
        if not only or case[0] in only:
            model_case(*case[:6], scale=case[6])
    if not only or "p19_traj20" in only:
        traj_case()
