"""K1 forward / backward time by hipGraph replay (the measurement bench.py reports as roofline), standalone and quick:
    python tools/k1_time.py [B]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raindrop_amd import synth
from raindrop_amd.models_rd import Raindrop_v2
dev = torch.device("cuda")
cfg = synth.make_config("P19"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(1)
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                synth.make_structure(cfg, "ones")).to(dev).train()
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, B, seed=100).items()}
r = bench.k1_roofline(m, cfg, b)
print("K1 fwd %.2f us  bwd %.2f us  total %.2f us  frac %.4f" % (r["fwd_us"], r["bwd_us"], r["fwd_us"] + r["bwd_us"], r["frac"]))
