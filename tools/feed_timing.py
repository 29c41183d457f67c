"""Batch feed: rd_batch_gather per batch (hipGraph replays, HIP events) against its HBM
roofline, beside the reference-style host slice + H2D (code/Raindrop.py:310-317), P19 scale (N=38 803)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raindrop_amd import feed, synth
cfg = synth.make_config("P19"); N = 38803; B = 256
b = synth.make_batch(cfg, 2048, seed=1)
rep = (N + 2047) // 2048
P = b["src"].repeat(1, rep, 1)[:, :N].contiguous(); Tm = b["times"].repeat(1, rep)[:, :N].contiguous()
S = b["static"].repeat(rep, 1)[:N].contiguous(); y = b["y"].repeat(rep)[:N].contiguous()
ds = feed.DeviceDataset(P, Tm, S, y)
idx = torch.from_numpy(np.random.default_rng(0).integers(0, N, B)).cuda()
out = ds.alloc(B)
for _ in range(5): ds.batch(idx, out=out)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): ds.batch(idx, out=out)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(20): ds.batch(idx, out=out)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); [g.replay() for _ in range(10)]; e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 200 * 1e3
T, W = cfg["max_len"], 2 * cfg["d_inp"]
byt = 2 * B * T * (W + 1) * 4 + B * (T * 4 + 2 * cfg["d_static"] * 4 + 24)
print("rd_batch_gather P19 N=%d B=%d: %.2f us per batch (hipGraph replays), %.1f MB -> %.0f GB/s = %.1f %% of 8 TB/s" % (
    N, B, us, byt / 1e6, byt / us / 1e3, byt / us / 1e3 / 80))
ih = idx.cpu().numpy()
t0 = time.perf_counter()
for _ in range(20):
    a = P[:, ih, :].cuda(); bb = Tm[:, ih].cuda(); c = S[ih].cuda(); d = y[ih].cuda(); l = torch.sum(bb > 0, dim=0)
torch.cuda.synchronize()
print("reference-style host slice + H2D (%d host threads): %.0f us per batch" % (torch.get_num_threads(), (time.perf_counter() - t0) / 20 * 1e6))
