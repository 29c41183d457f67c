"""CPU, world_size 2 (gloo): the flat-gradient bucketed all-reduce gives every rank the gradient of
the FULL batch when each rank back-propagates the CE-mean of its own equal shard.  The model
function is the oracle's restatement (the HIP path needs a GPU); the DP layer under test is the
product code in raindrop_amd/dp.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import restatement as O2
from raindrop_amd import dp, synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_buckets, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = synth.make_config("TINY")
    gs = synth.make_structure(cfg, "sparse")
    import json
    surf = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_surface.json")))["TINY"]
    names = synth.live_parameter_names(cfg)
    params = [(n, torch.nn.Parameter(synth.param_values(n, surf[n], seed=rank * 17))) for n in names]  # differ per rank
    holder = torch.nn.ParameterList([p for _, p in params])
    dp.broadcast_parameters(holder, src=0)                       # ... until rank 0's are broadcast
    flat = dp.FlatGradAllReduce(params, n_buckets=n_buckets)
    full = synth.make_batch(cfg, 8, seed=2)
    shard = dp.shard_batch(full, rank, world)
    p = dict(params)
    for _ in range(2):                                           # two steps: zero() must reset state
        flat.zero()
        logits, _ = O2.raindrop_v2_forward(p, cfg, shard["src"], shard["static"], shard["times"],
                                           shard["lengths"], gs)
        torch.nn.functional.cross_entropy(logits, shard["y"]).backward()
        flat.finish()
    got = {n: t.grad.clone() for n, t in params}
    # single-process reference on the full batch with rank 0's parameters
    p0 = {n: synth.param_values(n, surf[n], seed=0).requires_grad_(True) for n in names}
    _, _, ref = O2.step_fwd_bwd(p0, cfg, full, gs)
    err = max(float((got[n] - ref[n]).abs().max() / (ref[n].abs().max() + 1e-30)) for n in names)
    ret[rank] = (err, all(t.grad.data_ptr() >= flat.flat.data_ptr() for _, t in params), flat.n_buckets)
    dist.destroy_process_group()


@pytest.mark.parametrize("n_buckets", [1, 3])
def test_flat_grad_allreduce_world2(n_buckets):
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_buckets, ret), nprocs=world, join=True)
    for rank in range(world):
        err, in_flat, nb = ret[rank]
        assert err < 1e-5, (rank, err)
        assert in_flat and nb == n_buckets


def test_shard_batch_is_a_partition():
    cfg = synth.make_config("TINY")
    full = synth.make_batch(cfg, 8, seed=0)
    parts = [dp.shard_batch(full, r, 4) for r in range(4)]
    assert torch.equal(torch.cat([p["src"] for p in parts], dim=1), full["src"])
    assert torch.equal(torch.cat([p["lengths"] for p in parts]), full["lengths"])
    with pytest.raises(AssertionError):
        dp.shard_batch(full, 0, 3)


def test_flattened_adam_equals_per_tensor_adam():
    """One Adam over the flat parameter buffer == Adam over the individual tensors (elementwise)."""
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5))]
    b = [torch.nn.Parameter(t.detach().clone()) for t in a]
    flat = dp.FlatGradAllReduce([("w", b[0]), ("v", b[1])], n_buckets=1)
    opt_a = torch.optim.Adam(a, lr=1e-2)
    opt_b = torch.optim.Adam([flat.flatten_parameters()], lr=1e-2)
    for step in range(3):
        for params, opt, zero, fin in ((a, opt_a, lambda: opt_a.zero_grad(), lambda: None),
                                       (b, opt_b, flat.zero, flat.finish)):
            zero()
            loss = (params[0] ** 2).sum() * (step + 1) + (params[1] * 3).sum()
            loss.backward()
            fin()
            opt.step()
    assert torch.allclose(a[0], b[0], atol=1e-7) and torch.allclose(a[1], b[1], atol=1e-7)
    assert b[0].data_ptr() == flat.flat_param.data_ptr()          # parameters live in the flat buffer
