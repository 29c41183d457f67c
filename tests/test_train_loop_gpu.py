"""GPU: the loop body of the reference's training script (`code/Raindrop.py:290-324`: balanced batch plan, batch slice,
`lengths`, `model.forward`, CrossEntropyLoss, backward, Adam) and its validation (`utils_rd.evaluate_standard`) driven
through the device-resident feed with the HIP model -- one epoch on a synthetic P19-shaped training split, then a
checkpoint round trip (`state_dict` / `load_state_dict`, `:374,381`).  The unmodified script itself is run end to end
on CPU by tests/test_compat_e2e.py (the GPU box has no reference tree)."""
import io

import numpy as np
import pytest
import torch

from raindrop_amd import feed, synth
from tests.helpers import build_ours

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_one_epoch_of_the_scripts_loop_body():
    cfg = synth.make_config("P19")
    N = 640
    data = synth.make_batch(cfg, N, seed=71)
    rng = np.random.default_rng(5)
    # labels correlated with one input channel so that one epoch can learn something
    ytrain = ((data["src"][:, :, 0].sum(0).numpy() + 0.5 * rng.standard_normal(N)) > 0).astype(np.int64)
    if ytrain.mean() > 0.5:
        ytrain = 1 - ytrain                                              # class 1 is the minority the plan triples
    ds = feed.DeviceDataset(data["src"], data["times"], data["static"], torch.from_numpy(ytrain), device=DEV)
    val = synth.make_batch(cfg, 96, seed=72)
    vds = feed.DeviceDataset(val["src"], val["times"], val["static"], None, device=DEV)
    torch.manual_seed(1)                                                 # Raindrop.py:58
    np.random.seed(0)
    model = build_ours(cfg, synth.make_structure(cfg, "ones"), DEV, 3)
    model.dropout.p = 0.2
    criterion = torch.nn.CrossEntropyLoss()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)                 # the script's optimizer, unchanged (:256)
    model.train()
    plan = feed.epoch_index_plan(ytrain, batch_size=128, strategy=2)     # :262-307
    assert len(plan) >= 2
    losses = []
    for idx in plan:
        P, Pstatic, Ptime, y, lengths = ds.batch(idx)                    # :310-317 in one launch
        outputs, reg, _ = model.forward(P, Pstatic, Ptime, lengths)      # :319
        opt.zero_grad()
        loss = criterion(outputs, y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and float(reg) == 0.0
    out_val = feed.evaluate_chunked(model, vds, chunk=40)                # utils_rd.py:310-320, chunked
    assert tuple(out_val.shape) == (96, 2) and bool(torch.isfinite(out_val).all())
    assert torch.equal(out_val, feed.evaluate_sharded(model, vds, chunk=40))          # no process group: same path
    # checkpoint round trip: a fresh model loaded from the saved state_dict evaluates identically
    buf = io.BytesIO()
    torch.save(model.state_dict(), buf)
    buf.seek(0)
    m2 = build_ours(cfg, synth.make_structure(cfg, "ones"), DEV, 99)
    m2.load_state_dict(torch.load(buf))
    assert torch.equal(feed.evaluate_chunked(m2, vds, chunk=96), feed.evaluate_chunked(model, vds, chunk=96))
