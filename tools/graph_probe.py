"""Debug: which part of the K1 path survives hipGraph capture (run on the GPU box)."""
import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, ops, synth
from raindrop_amd.models_rd import Raindrop_v2
dev = torch.device("cuda")
cfg = synth.make_config("P19"); B = 256
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                synth.make_structure(cfg, "ones")).to(dev)
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, B, seed=0).items()}
g = m._graph(dev); shp = _lib.shape(B, 60, 34, 4)
args = (b["src"], b["times"], b["lengths"], m.pos_encoder.timescales(dev), g["ssum"], m.R_u,
        m.ob_propagation.lin_value.weight, m.ob_propagation.lin_value.bias,
        m.ob_propagation_layer2.lin_value.weight, m.ob_propagation_layer2.lin_value.bias, shp, 0.2, 1)
dz = torch.randn(60, B, 152, device=dev)
wrt = [m.R_u, m.ob_propagation.lin_value.weight]
def fwd_only():
    with torch.no_grad(): ops.sensor_stage(*args)
def fwd_bwd():
    z, _ = ops.sensor_stage(*args); torch.autograd.grad(z, wrt, dz)
for name, fn in (("fwd_only", fwd_only), ("fwd_bwd", fwd_bwd)):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    print(name, "warm ok", flush=True)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    print(name, "captured", flush=True)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(name, "replay us", e0.elapsed_time(e1) / 50 * 1e3, flush=True)
