"""Debug: phase stamps (clock64, thread 0 of workgroup 0) of k_head_rows (rd_head.hip) inside a real token-plan training step.
Usage: head_timing.py [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, dp, synth
from raindrop_amd.models_rd import Raindrop_v2
from raindrop_amd.step import TrainStep
lib = _lib.load()
lib.rd_debug_set_head_stamps.argtypes = [ctypes.c_void_p]
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
stamps = torch.zeros(32, dtype=torch.int64, device=dev)
lib.rd_debug_set_head_stamps(stamps.data_ptr())
ts.run()
torch.cuda.synchronize()
lib.rd_debug_set_head_stamps(None)
s = stamps.cpu().tolist()
N = ["start", "small operands, rows, W0 requested", "masked-mean rows summed", "barrier", "mean reduced + emb; barrier", "hid (W0 feat); barrier",
     "logits; barrier", "softmax / loss; barrier", "dhid + workspace rows; barrier", "dfeat partials; barrier", "dfeat reduced; barrier", "dr rows out"]
t0, prev = s[0], s[0]
for i, name in enumerate(N):
    print("  %-40s %7d  +%6d" % (name, s[i] - t0, s[i] - prev)); prev = s[i]
