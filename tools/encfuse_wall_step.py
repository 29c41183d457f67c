"""Debug: wall-clock stamps of every workgroup of k_enc_post_fwd inside a TrainStep (eager enqueue, token plan on)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, dp, synth
from raindrop_amd.models_rd import Raindrop_v2
from raindrop_amd.step import TrainStep
lib = _lib.load()
lib.rd_debug_set_encfuse_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
cfg = synth.make_config("P19")
B = 256
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                synth.make_structure(cfg, "ones")).to(dev).train()
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, B, seed=100).items()}
named = dict(m.named_parameters())
flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
ts = TrainStep(m, flat, b, use_graph=False, autotune=False)
for _ in range(3):
    ts.run()
torch.cuda.synchronize()
stamps = torch.zeros(256 + 2048, dtype=torch.int64, device=dev)
lib.rd_debug_set_encfuse_stamps(stamps.data_ptr())
ts.run()
torch.cuda.synchronize()
lib.rd_debug_set_encfuse_stamps(None)
s = stamps.cpu()
n = 480
st = s[256:256 + 2 * n].view(n, 2); en = s[256 + 1024:256 + 1024 + 2 * n].view(n, 2)
live = (st[:, 0] != 0)
print("plan[0..2] =", ts.plan[:3].tolist() if ts.plan is not None else None, " workgroups that ran:", int(live.sum()))
st, en = st[live], en[live]
w0 = int(st[:, 0].min())
dur_w = (en[:, 0] - st[:, 0]).double() * 10e-3
print("first start -> last end: %.1f us; start skew %.1f us; per-WG us min %.1f median %.1f max %.1f" % (
    (int(en[:, 0].max()) - w0) * 10e-3, (int(st[:, 0].max()) - w0) * 10e-3, dur_w.min(), dur_w.median(), dur_w.max()))
