"""Host time per step of the three step forms on ONE GPU with the collectives issued for real (a one-rank `nccl` group,
FlatGradAllReduce(force_collective=True)): what the Python side of an N > 1 rank spends per step.
    (a) two hipGraphs + two dist.all_reduce + two waits + FlatAdam.step         (rounds 3-5 default at N > 1)
    (b) TrainStep.capture_full: ONE replay                                        (round 6 default at every N)
python tools/host_side_timing.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from raindrop_amd import dp, synth
from raindrop_amd.models_rd import Raindrop_v2
from raindrop_amd.optim import FlatAdam
from raindrop_amd.step import TrainStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
cfg = synth.make_config("P19")
for full in (False, True):
    torch.manual_seed(1)
    m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                    synth.make_structure(cfg, "ones")).to(dev).train()
    b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, 256, seed=100).items()}
    named = dict(m.named_parameters())
    flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2, force_collective=True)
    opt = FlatAdam(flat.flatten_parameters(), lr=1e-4)
    ts = TrainStep(m, flat, b, split=True, autotune=False)
    if full:
        ts.capture_full(opt)
    one = (lambda: ts.run_full()) if full else (lambda: (ts.run_allreduce(), opt.step()))
    for _ in range(5):
        one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        one()
    th = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("%-58s ms/step %.4f   host us/step %.1f   loss %.6f" % ("ONE hipGraph per step (capture_full, collectives captured)" if full else
          "two graphs + 2 x dist.all_reduce + waits + Adam launch", (t1 - t0) * 1e3 / steps, (th - t0) * 1e6 / steps, float(ts.loss)))
    ts.close()
dist.destroy_process_group()
