"""Time the hipGraph training step (TrainStep.run + flat Adam) of any BASELINE config:   python tools/step_cfg.py CONFIG B STEPS
(P12 / PAM / SYN256 counterpart of tools/step_only.py; target of rocprofv3 --kernel-trace for those shapes).  RD_PRECISION and
the RD_* switches of README.md apply."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from raindrop_amd import dp, synth
from raindrop_amd.models_rd import Raindrop_v2
from raindrop_amd.optim import FlatAdam
from raindrop_amd.step import TrainStep
cfgn, B = sys.argv[1], int(sys.argv[2]); steps = int(sys.argv[3])
dev = torch.device("cuda"); cfg = synth.make_config(cfgn); torch.manual_seed(1)
kw = {} if cfg["static"] else {"static": False}
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", cfg["n_classes"],
                synth.make_structure(cfg, "ones"), **kw).to(dev).train()
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, B, seed=100).items()}
named = dict(m.named_parameters())
flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
opt = FlatAdam(flat.flatten_parameters(), lr=1e-4)
ts = TrainStep(m, flat, b, autotune=False)
for _ in range(3): ts.run(); opt.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): ts.run(); opt.step()
torch.cuda.synchronize(); print("ms/step %.4f" % ((time.perf_counter() - t0) * 1e3 / steps))
