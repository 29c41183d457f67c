"""Run the hipGraph training step bench.py times (TrainStep.run + flat Adam), nothing else: the target for
rocprofv3 --kernel-trace of the GRAPH step (not the eager autograd step).   python tools/step_only.py [steps] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import dp, synth
from raindrop_amd.models_rd import Raindrop_v2
from raindrop_amd.optim import FlatAdam
from raindrop_amd.step import TrainStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda")
cfg = synth.make_config("P19")
torch.manual_seed(1)
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                synth.make_structure(cfg, "ones")).to(dev).train()
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, B, seed=100).items()}
named = dict(m.named_parameters())
flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
opt = FlatAdam(flat.flatten_parameters(), lr=1e-4)
ts = TrainStep(m, flat, b, split=True if os.environ.get('RD_SPLIT') == '1' else None)   # RD_SPLIT=1: the two-graph form of N > 1 (cost of the split)
import time
full = os.environ.get("RD_FULL") == "1"          # RD_FULL=1: the whole step incl. all-reduce and Adam as ONE hipGraph (TrainStep.capture_full)
if full:
    ts.capture_full(opt)
one = (lambda: ts.run_full()) if full else (lambda: (ts.run_allreduce(), opt.step()))
ts.seed_cell.zero_()                              # the same dropout mask sequence whatever the capture's warm-up replays did
for _ in range(3):
    one()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    one()
th = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
print("loss", float(ts.loss), "ms/step %.4f" % ((t1 - t0) * 1e3 / steps), "host us/step %.1f" % ((th - t0) * 1e6 / steps), "(one graph per step)" if full else "")
