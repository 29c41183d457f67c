"""Debug: phase stamps (clock64, thread 0 of workgroup 0 = the LONGEST sample of the batch) of the fused attention kernels
(rd_attnfuse.hip) inside a real token-plan training step.  Usage: attnfuse_timing.py [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, dp, synth
from raindrop_amd.models_rd import Raindrop_v2
from raindrop_amd.step import TrainStep
lib = _lib.load()
lib.rd_debug_set_attnfuse_stamps.argtypes = [ctypes.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
cfg = synth.make_config("P19")
torch.manual_seed(1)
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                synth.make_structure(cfg, "ones")).to(dev).train()
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, B, seed=100).items()}
named = dict(m.named_parameters())
flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
ts = TrainStep(m, flat, b, use_graph=False, autotune=False)
for _ in range(3):
    ts.run()
torch.cuda.synchronize()
stamps = torch.zeros(128, dtype=torch.int64, device=dev)
lib.rd_debug_set_attnfuse_stamps(stamps.data_ptr())
ts.run()                      # the LAST layer's launches overwrite the first layer's stamps: what is shown is layer 1 fwd, layer 0 bwd
torch.cuda.synchronize()
lib.rd_debug_set_attnfuse_stamps(None)
s = stamps.cpu().tolist()
FW = ["x requested", "x stored", "barrier"] + ["h%d %s" % (h, n) for h in range(2) for n in
      ("qkv projected", "barrier", "S + row max", "barrier; P stored", "barrier", "O = PV staged", "barrier", "rows out + barrier")]
print("forward (cycles since the first stamp; delta)")
t0, prev = s[0], s[0]
for i, name in enumerate(FW):
    if s[i]:
        print("  %-24s %7d  +%6d" % (name, s[i] - t0, s[i] - prev)); prev = s[i]
BW = ["requests issued", "x stored", "barrier A", "dO stored", "qkv projected", "barrier B", "S, dP, P, dS stored", "barrier C", "dQ dK dV",
      "barrier D", "planes stored", "barrier E", "dqkv tiles out", "dx product"]
print("backward")
t0, prev = s[32], s[32]
for h in range(2):
    for i, name in enumerate(BW):
        v = s[32 + 16 * h + i]
        if v:
            print("  h%d %-22s %7d  +%6d" % (h, name, v - t0, v - prev)); prev = v
for i, name in ((64, "final barrier"), (65, "dx rows out")):
    if s[i]:
        print("  %-25s %7d  +%6d" % (name, s[i] - t0, s[i] - prev)); prev = s[i]
