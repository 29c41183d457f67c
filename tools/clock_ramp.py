"""Does the step get faster with sustained load (clock ramp / power state)?  Windows of 1000 graph steps back to back, ms/step per
window, rocm-smi clocks sampled beside it.   python tools/clock_ramp.py [windows] [steps_per_window]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import dp, synth
from raindrop_amd.models_rd import Raindrop_v2
from raindrop_amd.optim import FlatAdam
from raindrop_amd.step import TrainStep
W = int(sys.argv[1]) if len(sys.argv) > 1 else 30
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1000


def smi(tag):
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in o.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "socclk", "Power", "Performance Level", "Temperature (Sensor junction)"))]
        print(tag, " | ".join(keep), flush=True)
    except Exception as ex:
        print(tag, "rocm-smi failed", ex, flush=True)


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
smi("idle  ")
for w in range(W):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(S):
        ts.run_allreduce(); opt.step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("window %2d  %.4f ms/step" % (w, (t1 - t0) * 1e3 / S), flush=True)
    if w in (0, W // 2, W - 1):
        smi("after %2d" % w)
