"""GPU, world_size 2 on ONE device (gloo): the data-parallel training step as bench.py runs it at N > 1 --
`TrainStep` (HIP forward + CE + backward into the flat gradient buffer) -> `flat.allreduce()` -> `FlatAdam` --
gives every rank the parameters a single process gets from the full batch.  An 8-GPU node is not available to
this repo's tests; the collective here is gloo on two processes that share cuda:0, the product code path
(raindrop_amd/step.py, dp.py, optim.py and every kernel) is the one bench.py --gpus N uses with RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

STEPS = 2
B_GLOBAL = 16


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run_steps(batch, world_rank=None):
    """STEPS training steps on `batch`; returns (flat all-reduced gradient of the first step, flat parameters after the last)."""
    from raindrop_amd import dp, synth
    from raindrop_amd.models_rd import Raindrop_v2
    from raindrop_amd.optim import FlatAdam
    from raindrop_amd.step import TrainStep
    dev = torch.device("cuda", 0)
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "sparse")
    m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], cfg["nhead"], cfg["nhid"], cfg["nlayers"], cfg["dropout"],
                    cfg["max_len"], cfg["d_static"], cfg["MAX"], 0.5, cfg["aggreg"], cfg["n_classes"], gs,
                    sensor_wise_mask=False)
    synth.fill_params_(m, seed=21)
    m = m.to(dev).train()
    named = dict(m.named_parameters())
    flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
    opt = FlatAdam(flat.flatten_parameters(), lr=1e-3)
    b = {k: (None if v is None else v.to(dev)) for k, v in batch.items()}
    ts = TrainStep(m, flat, b, p_drop=0.0, use_graph=False)
    g_first = None
    for i in range(STEPS):
        ts.run()
        flat.allreduce()
        if i == 0:
            g_first = flat.flat.detach().cpu().numpy().copy()
        opt.step()
    torch.cuda.synchronize()
    ts.close()
    return g_first, flat.flat_param.detach().cpu().numpy().copy()


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raindrop_amd import dp, synth
    full = synth.make_batch(synth.make_config("P19"), B_GLOBAL, seed=33)
    g, p = _run_steps(dp.shard_batch(full, rank, world))
    ret[rank] = (g, p)
    dist.destroy_process_group()


def test_two_rank_step_equals_full_batch_step():
    from raindrop_amd import synth
    full = synth.make_batch(synth.make_config("P19"), B_GLOBAL, seed=33)
    g_ref, p_ref = _run_steps(full)                               # this process: no process group, world == 1
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    g0, p0 = ret[0]
    g1, p1 = ret[1]
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1)      # replicas stay bit-identical
    gscale = np.abs(g_ref).max()
    assert np.abs(g0 - g_ref).max() <= 3e-5 * gscale, np.abs(g0 - g_ref).max() / gscale
    # Adam's update is lr * m / (sqrt(v) + eps): a 1e-5 relative gradient difference moves a weight by ~1e-5 * lr.  Adam
    # also normalises away the magnitude, so an entry whose gradient is pure summation noise (|g| far below 1e-5 of the
    # largest) can take a different sign and move by up to 2 lr per step: those may exist, but only as a tiny fraction.
    dp_ = np.abs(p0 - p_ref)
    # (measured 0.05-0.3 % of the 505 k entries, depending on how the kernels group their partial sums at B and B/2)
    assert (dp_ > 1e-3 * 1e-3 * STEPS).mean() < 1e-2, float((dp_ > 2e-6).mean())
    assert dp_.max() <= 2.1 * 1e-3 * STEPS


RANK_SEED_STRIDE = 0x9E3779B97F4A7C15          # raindrop_amd.ops.rank_seed_offset


def _graph_dropout_step(batch, seed, overlapped=False):
    """ONE hipGraph step with dropout 0.2 (the configuration bench.py --gpus N runs): returns (loss, flat gradient after
    flat.allreduce())."""
    from raindrop_amd import dp, synth
    from raindrop_amd.models_rd import Raindrop_v2
    from raindrop_amd.step import TrainStep
    dev = torch.device("cuda", 0)
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "sparse")
    m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], cfg["nhead"], cfg["nhid"], cfg["nlayers"], cfg["dropout"],
                    cfg["max_len"], cfg["d_static"], cfg["MAX"], 0.5, cfg["aggreg"], cfg["n_classes"], gs,
                    sensor_wise_mask=False)
    synth.fill_params_(m, seed=21)
    m = m.to(dev).train()
    named = dict(m.named_parameters())
    flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
    b = {k: (None if v is None else v.to(dev)) for k, v in batch.items()}
    ts = TrainStep(m, flat, b, p_drop=0.2, use_graph=True, seed=seed, autotune=False)
    ts.seed_cell.zero_()                                          # the capture's warm-up runs bumped it: one defined replay
    if overlapped:                                                # two graphs, the tail bucket's collective started between them
        assert ts.split and ts.graph_b is not None and 0 < ts.early_grad_offset() < flat.flat.numel()
        loss = float(ts.run_allreduce())
    else:
        loss = float(ts.run())
        flat.allreduce()
    torch.cuda.synchronize()
    g = flat.flat.detach().cpu().numpy().copy()
    ts.close()
    return loss, g


def _graph_worker(rank, world, port, ret, overlapped=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raindrop_amd import dp, synth
    full = synth.make_batch(synth.make_config("P19"), B_GLOBAL, seed=33)
    ret[rank] = _graph_dropout_step(dp.shard_batch(full, rank, world), seed=1234, overlapped=overlapped)
    dist.destroy_process_group()


@pytest.mark.parametrize("overlapped", [False, True])
def test_two_rank_graph_step_with_dropout(overlapped):
    """The step bench.py --gpus N actually runs -- hipGraph replay, dropout 0.2, per-rank seed offsets, device seed cell -- on two
    ranks: the all-reduced gradient is bit-identical on both ranks, the ranks drew DIFFERENT masks (different local losses), and
    the result equals a single process replaying the two shards with the two ranks' seeds and averaging.  `overlapped`: the
    split form bench.py uses at N > 1 (TrainStep.run_allreduce: two graphs, the collective of the last layer's + head's
    gradients started between them) against the same single-process one-graph replay: bit-equal."""
    from raindrop_amd import dp, synth
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_graph_worker, args=(world, port, ret, overlapped), nprocs=world, join=True)
    (l0, g0), (l1, g1) = ret[0], ret[1]
    assert np.array_equal(g0, g1)
    assert l0 != l1
    full = synth.make_batch(synth.make_config("P19"), B_GLOBAL, seed=33)
    rep = []
    for r in range(world):                                        # no process group here: rank_seed_offset() == 0, so pass the rank's seed
        seed = (1234 + r * RANK_SEED_STRIDE) & 0x7FFFFFFFFFFFFFFF
        rep.append(_graph_dropout_step(dp.shard_batch(full, r, world), seed=seed))
    assert rep[0][0] == l0 and rep[1][0] == l1                    # same masks as the ranks drew
    avg = (rep[0][1] + rep[1][1]) / np.float32(world)
    assert np.array_equal(avg, g0)


def _eval_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ret[rank] = _eval_logits(sharded=True)
    dist.destroy_process_group()


def _eval_logits(sharded):
    from raindrop_amd import feed, synth
    from tests.helpers import build_ours
    cfg = synth.make_config("P19")
    val = synth.make_batch(cfg, 101, seed=90)                              # odd size: the last shard is shorter
    ds = feed.DeviceDataset(val["src"], val["times"], val["static"], None, device="cuda:0")
    m = build_ours(cfg, synth.make_structure(cfg, "sparse"), "cuda:0", 5).eval()
    out = feed.evaluate_sharded(m, ds, chunk=32) if sharded else feed.evaluate_chunked(m, ds, chunk=32)
    return out.cpu().numpy().copy()


def test_sharded_evaluate_standard_all_gathers_the_same_logits():
    """`utils_rd.evaluate_standard` sharded over 2 ranks + logits all-gather == the single-process result, bit for bit."""
    ref = _eval_logits(sharded=False)
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_eval_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0].shape == (101, 2)
    assert np.array_equal(ret[0], ref) and np.array_equal(ret[1], ref)


def test_bench_two_ranks_launch_line_on_one_gpu():
    """`bench.py --gpus 2` launched EXACTLY as the driver launches it (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 ...`), with RD_BENCH_ONE_GPU=1 so that both ranks share cuda:0 and talk gloo:
    the whole N > 1 control flow of the benchmark -- process-group setup, the collective step-mode decision, per-rank batches,
    broadcast of the parameters, barriers around the timed region, MAX over ranks, rank 0's single JSON line -- runs on every
    1-GPU box.  (The collective itself is RCCL on a real node; no scaling curve has been measured by this repo.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RD_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "32", "--no-roofline", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=root)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, (res.returncode, res.stdout[-2000:], res.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "samples/s"
    assert d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 2 * 32 * 3 / (d["ms_per_step"] * 3e-3)) <= 0.02 * d["value"]
    assert "all-reduce" in d["config"]["workload"]


def test_bench_whole_step_graph_probe_child_and_its_failure_path():
    """Guard (0) of the whole-step graph at N > 1 (bench.full_graph_probe_ok): every rank's CHILD tries capture + replay in a
    process group of its own, so that a hang costs the graph, not the line.  (a) The child itself (`--full-graph-probe`, here as a
    single process) captures the whole step, replays it and reports a finite loss.  (b) Two ranks as the driver launches them, on
    one GPU over gloo with the probe forced: the children get a fresh port, form their group, FAIL (gloo collectives cannot be
    captured), every rank reports it, and the parents still print the one JSON line with the two-graph step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--full-graph-probe", "--gpus", "1", "--batch", "32",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert res.returncode == 0 and any(ln.startswith("FULLPROBE ok") for ln in res.stdout.splitlines()), (res.stdout[-1000:], res.stderr[-2000:])
    env.update(RD_BENCH_ONE_GPU="1", RD_BENCH_FORCE_FULL_PROBE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "32", "--no-roofline", "--no-cpu-baseline"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and len(lines) == 1, (res.returncode, res.stdout[-2000:], res.stderr[-2000:])
    assert res.stderr.count("whole-step graph probe failed") == 2, res.stderr[-3000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "ONE hipGraph per step" not in str(d["config"].get("step_mode"))


def _rccl_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from raindrop_amd import dp, synth
    from raindrop_amd.models_rd import Raindrop_v2
    from raindrop_amd.step import TrainStep
    dev = torch.device("cuda", 0)
    cfg = synth.make_config("P19")
    out = []
    for split in (False, True):
        m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], cfg["nhead"], cfg["nhid"], cfg["nlayers"], cfg["dropout"], cfg["max_len"],
                        cfg["d_static"], cfg["MAX"], 0.5, cfg["aggreg"], cfg["n_classes"], synth.make_structure(cfg, "sparse"),
                        sensor_wise_mask=False)
        synth.fill_params_(m, seed=21)
        m = m.to(dev).train()
        named = dict(m.named_parameters())
        flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2, force_collective=split)
        b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, 16, seed=33).items()}
        ts = TrainStep(m, flat, b, p_drop=0.2, use_graph=True, seed=77, autotune=False, split=split)
        ts.seed_cell.zero_()
        for _ in range(3):                                          # three replays: the collectives interleave with graph replays
            loss = float(ts.run_allreduce())
        torch.cuda.synchronize()
        out.append((loss, flat.flat.detach().cpu().numpy().copy()))
        ts.close()
    ret[0] = out
    dist.destroy_process_group()


def test_rccl_one_rank_collectives_between_graph_replays():
    """RCCL itself on the one GPU this box has: a one-rank `nccl` process group, the two-graph step with its two asynchronous AVG
    all-reduces issued for real (force_collective), three replays -- loss and gradients bit-equal to the one-graph step without any
    collective.  (What N > 1 adds to this is other ranks' data; backend load, reduce op, stream ordering against hipGraph replays and
    the handle protocol are all exercised here.)"""
    world, port = 1, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_worker, args=(world, port, ret), nprocs=1, join=True)
    (l0, g0), (l1, g1) = ret[0]
    assert l0 == l1 and np.array_equal(g0, g1)


def _rccl_full_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from raindrop_amd import dp, synth
    from raindrop_amd.models_rd import Raindrop_v2
    from raindrop_amd.optim import FlatAdam
    from raindrop_amd.step import TrainStep
    dev = torch.device("cuda", 0)
    cfg = synth.make_config("P19")
    out = []
    for full in (False, True):
        m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], cfg["nhead"], cfg["nhid"], cfg["nlayers"], cfg["dropout"], cfg["max_len"],
                        cfg["d_static"], cfg["MAX"], 0.5, cfg["aggreg"], cfg["n_classes"], synth.make_structure(cfg, "sparse"),
                        sensor_wise_mask=False)
        synth.fill_params_(m, seed=21)
        m = m.to(dev).train()
        named = dict(m.named_parameters())
        flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2, force_collective=True)
        opt = FlatAdam(flat.flatten_parameters(), lr=1e-3)
        b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, 16, seed=33).items()}
        ts = TrainStep(m, flat, b, p_drop=0.2, use_graph=True, seed=77, autotune=False, split=True)
        if full:
            ts.capture_full(opt)
        ts.seed_cell.zero_()
        losses = []
        for i in range(6):
            if i == 4:
                opt.lr = 2.5e-4                      # what ReduceLROnPlateau does (code/Raindrop.py:257-259): the captured step must follow
            if full:
                losses.append(float(ts.run_full()))
            else:
                losses.append(float(ts.run_allreduce())); opt.step()
        torch.cuda.synchronize()
        out.append((losses, opt.t, flat.flat.detach().cpu().numpy().copy(), opt.param.detach().cpu().numpy().copy()))
        ts.close()
        del ts
    ret[0] = out
    dist.destroy_process_group()


def test_rccl_one_rank_whole_step_as_one_graph():
    """TrainStep.capture_full: forward + loss + backward, BOTH asynchronous AVG all-reduces (one-rank `nccl` group, issued for real:
    RCCL under stream capture, the collectives' stream forked from and joined to the captured one) and the optimizer (device step
    cell + rd_adam_step_dev) as ONE hipGraph; six replays with dropout on (the learning rate lowered after the fourth: the step is
    captured again, the dropout stream keeps its place) against six eager-collective steps (two graphs, two
    dist.all_reduce calls, FlatAdam.step with host-computed bias corrections): the same losses, gradients bit-equal, parameters
    equal to rounding of the device-side bias corrections (double precision on both sides)."""
    world, port = 1, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_full_worker, args=(world, port, ret), nprocs=1, join=True)
    (l0, t0, g0, p0), (l1, t1, g1, p1) = ret[0]
    assert t0 == t1 == 6
    dp_, dg_ = np.abs(p0 - p1).max(), np.abs(g0 - g1).max()
    msg = "losses %r | %r, max |dp| %.3g, max |dg| %.3g of %.3g" % (l0, l1, dp_, dg_, np.abs(g0).max())
    assert np.allclose(l0, l1, rtol=1e-6, atol=1e-7), msg
    assert dg_ <= 1e-6 * max(np.abs(g0).max(), 1e-30), msg
    assert dp_ <= 4e-7 * max(1.0, np.abs(p0).max()), msg


def _tuned_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("RD_RG_ROWS32", None); os.environ.pop("RD_RG_WAVES16", None)      # pinned knobs switch the tuner off
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raindrop_amd import dp, synth
    from raindrop_amd.step import TrainStep
    from tests.helpers import build_ours
    dev = torch.device("cuda", 0)
    cfg = synth.make_config("P19")
    m = build_ours(cfg, synth.make_structure(cfg, "sparse"), dev, 21).train()
    named = dict(m.named_parameters())
    flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
    full = synth.make_batch(cfg, 64, seed=33)
    b = {k: (None if v is None else v.to(dev)) for k, v in dp.shard_batch(full, rank, world).items()}
    ts = TrainStep(m, flat, b, p_drop=0.0, use_graph=True, autotune=True)      # split form + tuner: both on by default at N > 1
    loss = float(ts.run_allreduce())
    torch.cuda.synchronize()
    ret[rank] = (ts.tuned_rows32, ts.tuned_waves16, ts.split, loss, flat.flat.detach().cpu().numpy().copy())
    ts.close()
    dist.destroy_process_group()


def test_two_ranks_tune_to_the_same_kernel_variants():
    """Under torch.distributed the capture-time tuner sums every variant's time over the ranks before choosing: both ranks must end
    up with the SAME (rows32, waves16) -- the step is then the same program on every rank and the one a single process would tune
    to -- and the all-reduced gradients are bit-identical."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_tuned_worker, args=(world, port, ret), nprocs=world, join=True)
    a, b = ret[0], ret[1]
    assert a[0] is not None and a[1] is not None and a[2] and b[2]
    assert (a[0], a[1]) == (b[0], b[1]), (a[:2], b[:2])
    assert a[0] in (15, 0, 3, 12) and a[1] in (12, 15)
    assert np.array_equal(a[4], b[4])


def test_train_step_refuses_what_it_does_not_implement():
    """TrainStep implements the default branch and (split form) relies on the forward order of the flat buffer: both are checked."""
    from raindrop_amd import _lib, dp, synth
    from raindrop_amd.step import TrainStep
    from tests.helpers import build_ours
    dev = torch.device("cuda", 0)
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "sparse")
    b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, 8, seed=3).items()}
    m = build_ours(cfg, gs, dev, 21, use_beta=True).train()
    named = dict(m.named_parameters())
    flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)])
    with pytest.raises(_lib.RaindropHipError, match="use_beta"):
        TrainStep(m, flat, b, use_graph=False)
    m = build_ours(cfg, gs, dev, 21).train()
    named = dict(m.named_parameters())
    live = set(synth.live_parameter_names(cfg))
    wrong = [(n, p) for n, p in m.named_parameters() if n in live]          # registration order: R_u, ob_propagation* behind the encoder
    flat = dp.FlatGradAllReduce(wrong)
    with pytest.raises(_lib.RaindropHipError, match="FORWARD order"):
        TrainStep(m, flat, b, use_graph=False, split=True)
    TrainStep(m, flat, b, use_graph=False, split=False).run()               # the one-graph form takes any order
    torch.cuda.synchronize()
