"""One captured P19 training step (B = 256, token plan, dropout on, seed cell zeroed): the loss and the flat gradient vector to a .npy,
then a few more steps with Adam, losses printed in full -- to compare two builds of the library (RD_LIB_PATH) step by step:
a difference in the first step is a different result, a difference that appears steps later is the amplification of rounding.
usage: step_dump.py <out.npy> [steps]       then: step_dump.py --compare a.npy b.npy"""
import os, sys
import numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a - b)
    print("first step: loss %r vs %r; gradient max |a - b| = %.3e of max |a| = %.3e (rel %.3e); entries that differ: %d of %d"
          % (float(a[0]), float(b[0]), d[1:].max(), np.abs(a[1:]).max(), d[1:].max() / np.abs(a[1:]).max(), int((d[1:] > 0).sum()), a.size - 1))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import dp, synth
from raindrop_amd.models_rd import Raindrop_v2
from raindrop_amd.optim import FlatAdam
from raindrop_amd.step import TrainStep
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda")
cfg = synth.make_config("P19")
torch.manual_seed(1)
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                synth.make_structure(cfg, "ones")).to(dev).train()
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, 256, seed=100).items()}
named = dict(m.named_parameters())
flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
opt = FlatAdam(flat.flatten_parameters(), lr=1e-4)
ts = TrainStep(m, flat, b, autotune=False)
ts.seed_cell.zero_()
ts.run()
torch.cuda.synchronize()
np.save(sys.argv[1], np.concatenate([[float(ts.loss)], flat.flat.detach().cpu().numpy().astype(np.float64)]))
losses = [float(ts.loss)]
for _ in range(steps - 1):
    opt.step(); ts.run(); losses.append(float(ts.loss))
torch.cuda.synchronize()
print("losses", " ".join("%.9g" % v for v in losses))
