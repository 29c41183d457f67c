"""GPU: trained-weight parity against the REFERENCE's own training loop (VERDICT r5 missing #3).

`tests/golden/p19_traj20.npz` is 20 optimizer steps of the reference model under the reference's loop body
(code/Raindrop.py:319-324: forward, zero_grad, CrossEntropyLoss, backward, Adam; lr 1e-3, dropout 0) on 20 different batches, made
by tests/golden/make_goldens.py from /root/reference.  Three questions, three tests:

1. Do OUR step and optimizer follow the reference's trajectory?  In the exact-fp32 arithmetic mode (same kernels, same code paths,
   fp32 MFMA) the 20 steps are replayed free-running four ways -- `TrainStep` + `FlatAdam`, `TrainStep.capture_full` (one hipGraph
   per step incl. the device-state Adam), and the nn.Module surface an unmodified script uses (operator by operator, and the
   captured module step) with torch's own Adam -- and every loss, the last logits, the held-out logits and the trained weights
   must agree (measured: losses to 1.2e-7).
2. At the reference's TRAINED weights (stored in full), does the default split-bf16 arithmetic still meet the north star?  Logits
   within 1e-4 in eval and train mode, loss 1e-5, every gradient within the bounds of tests/test_gpu_parity.py -- trained-weight
   parity proper, with the weights given.
3. How far does a free-running replay in split-bf16 drift?  Adam divides by sqrt(v): in the first steps the update of EVERY entry
   is ~lr * sign(g), so a gradient entry whose magnitude is below the arithmetic's error (~1e-3 of the matrix norm, from ReLU
   gates that flip at a 1e-5 forward difference; tools/grad_gate_diag.py) gets an O(lr) different update.  The drift is bounded
   here (loss trajectory within 2e-2, the first three steps within 2e-4) and documented in DESIGN.md -- it is a property of Adam on
   any perturbed gradient, not of these kernels: question 1 is the evidence, and tools/adam_noise_drift.py shows the fp32
   restatement itself drifting by 3e-3 in the loss / 0.08 in the held-out logits when 1e-6 relative noise is added to its
   gradients (profiles/r06_adam_noise_drift.txt)."""
import numpy as np
import pytest
import torch

from raindrop_amd import synth
from tests.helpers import build_ours, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(meta):
    cfg = synth.make_config(meta["cfg"])
    gs = synth.make_structure(cfg, meta["structure"])
    m = build_ours(cfg, gs, DEV, meta["param_seed"]).train()
    batches = [synth.make_batch(cfg, meta["batch"], seed=meta["batch_seed0"] + i) for i in range(meta["steps"])]
    held = synth.make_batch(cfg, meta["batch"], seed=meta["held_out_seed"])
    return cfg, gs, m, batches, held


def _held_logits(m, held):
    m.eval()
    hv = {k: (None if v is None else v.to(DEV)) for k, v in held.items()}
    with torch.no_grad():
        hl, _, _ = m(hv["src"], hv["static"], hv["times"], hv["lengths"])
    m.train()
    return hl.cpu().numpy()


def _weight_error(g, m):
    """worst error of a trained weight relative to HOW FAR TRAINING MOVED IT (against the weight's own norm an optimizer that never
    ran would pass)"""
    params = dict(m.named_parameters())
    worst = (0.0, "")
    for n in [str(x) for x in g["live"]]:
        err = float((params[n].detach().cpu().double() - torch.from_numpy(g["trained/" + n]).double()).norm()) / max(float(g["wmoved/" + n]), 1e-30)
        worst = max(worst, (err, n))
    return worst


def _run(path, m, cfg, batches, lr):
    """free-running replay of the 20 steps; returns (losses, logits of the last step's forward)"""
    from raindrop_amd import dp
    from raindrop_amd.optim import FlatAdam
    from raindrop_amd.step import TrainStep
    losses = []
    if path in ("step", "full"):
        named = dict(m.named_parameters())
        flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
        opt = FlatAdam(flat.flatten_parameters(), lr=lr)
        buf = {k: (None if v is None else v.to(DEV).clone()) for k, v in batches[0].items()}
        ts = TrainStep(m, flat, buf, p_drop=0.0, autotune=False, split=False)
        if path == "full":
            ts.capture_full(opt)
        for b in batches:
            for k, v in b.items():
                if v is not None:
                    buf[k].copy_(v)
            if path == "full":
                losses.append(float(ts.run_full()))
            else:
                losses.append(float(ts.run()))
                opt.step()
        last = ts.logits.detach().cpu().numpy().copy()
        ts.close()
        return losses, last
    m.graph_step = (path == "module_graph")
    opt = torch.optim.Adam(m.parameters(), lr=lr)                       # code/Raindrop.py:256
    crit = torch.nn.CrossEntropyLoss()
    for b in batches:
        dv = {k: (None if v is None else v.to(DEV)) for k, v in b.items()}
        outputs, _, _ = m.forward(dv["src"], dv["static"], dv["times"], dv["lengths"])      # :319
        opt.zero_grad()
        loss = crit(outputs, dv["y"])
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses, outputs.detach().cpu().numpy()


@pytest.mark.parametrize("path", ["step", "full", "module_operators", "module_graph"])
def test_reference_trajectory_in_exact_arithmetic(path):
    from raindrop_amd import _lib
    _lib.call("rd_set_precision", 0)
    try:
        g, meta = load_golden("p19_traj20")
        cfg, gs, m, batches, held = _setup(meta)
        losses, last = _run(path, m, cfg, batches, meta["lr"])
        dl = np.abs(np.array(losses) - g["losses"]).max()
        d_last, d_held = np.abs(last - g["last_logits"]).max(), np.abs(_held_logits(m, held) - g["held_logits"]).max()
        werr = _weight_error(g, m)
        print(path, "loss %.2e last logits %.2e held logits %.2e weights %.2e (%s)" % (dl, d_last, d_held, werr[0], werr[1]))
        assert dl < 2e-6 and d_last < 1e-5 and d_held < 1e-5, (path, dl, d_last, d_held)
        assert werr[0] < 2e-3, werr
    finally:
        _lib.call("rd_set_precision", 1)


@pytest.mark.parametrize("mode", ["bf16x3", "fp32"])
@pytest.mark.parametrize("graph_step", [False, True], ids=["operators", "module_graph"])
def test_parity_at_the_reference_s_trained_weights(mode, graph_step):
    from raindrop_amd import _lib
    from tests.test_gpu_parity import TOL, _grad_close_masked
    _lib.call("rd_set_precision", 1 if mode == "bf16x3" else 0)
    TOL["x"] = 6.0 if mode == "bf16x3" else 1.0
    try:
        g, meta = load_golden("p19_traj20")
        cfg, gs, m, batches, held = _setup(meta)
        params = dict(m.named_parameters())
        live = [str(x) for x in g["live"]]
        with torch.no_grad():
            for n in live:
                params[n].copy_(torch.from_numpy(g["trained/" + n]).to(DEV))
        assert np.abs(_held_logits(m, held) - g["held_logits"]).max() < 1e-4
        m.graph_step = graph_step
        hv = {k: (None if v is None else v.to(DEV)) for k, v in held.items()}
        logits, _, _ = m(hv["src"], hv["static"], hv["times"], hv["lengths"])
        loss = torch.nn.functional.cross_entropy(logits, hv["y"])
        loss.backward()
        assert np.abs(logits.detach().cpu().numpy() - g["held_logits_train"]).max() < 1e-4
        assert abs(float(loss) - float(g["held_loss"])) < 1e-5
        for n in live:
            st = int(g["heldgradstride/" + n])
            got = params[n].grad.detach().reshape(-1)[::st].cpu().numpy()
            _grad_close_masked(got, g["heldgrad/" + n], 1e-3, n)
            gn = float(g["heldgradnorm/" + n])
            assert abs(params[n].grad.double().norm().item() - gn) <= 2e-3 * gn + 1e-12, n
    finally:
        _lib.call("rd_set_precision", 1)
        TOL["x"] = 1.0


@pytest.mark.parametrize("path", ["full", "module_graph"])
def test_free_running_split_bf16_trajectory_stays_close(path):
    from raindrop_amd import _lib
    _lib.call("rd_set_precision", 1)
    g, meta = load_golden("p19_traj20")
    cfg, gs, m, batches, held = _setup(meta)
    losses, last = _run(path, m, cfg, batches, meta["lr"])
    d = np.abs(np.array(losses) - g["losses"])
    print(path, "loss drift per step", np.array2string(d, precision=1))
    assert d[:3].max() < 2e-4 and d.max() < 2e-2, d
    # (the held-out logits under the two sets of trained weights differ by ~0.2 -- as they do between two runs of the fp32
    # restatement whose gradients differ by 1e-6 relative Gaussian noise: tools/adam_noise_drift.py, profiles/r06_adam_noise_drift.txt)
    print(path, "held-out logits differ by %.3f" % np.abs(_held_logits(m, held) - g["held_logits"]).max())
