"""Debug: per-phase cycle stamps of the fused message-passing forward and backward kernels (first workgroups)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, ops, synth
from raindrop_amd.models_rd import Raindrop_v2
lib = _lib.load()
dev = torch.device("cuda")
cfg = synth.make_config("P19")
B = 256
gs = synth.make_structure(cfg, "ones")
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2, gs).to(dev)
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, B, seed=0).items()}
g = m._graph(dev)
shp = _lib.shape(B, 60, 34, 4)
stamps = torch.zeros(8 * 16, dtype=torch.int64, device=dev)
args = (b["src"], b["times"], b["lengths"], m.pos_encoder.timescales(dev), g["ssum"], m.R_u,
        m.ob_propagation.lin_value.weight, m.ob_propagation.lin_value.bias,
        m.ob_propagation_layer2.lin_value.weight, m.ob_propagation_layer2.lin_value.bias, shp, 0.2, 1)
for _ in range(3):
    ops.sensor_stage(*args)
lib.rd_debug_set_stamps.argtypes = [ctypes.c_void_p]
lib.rd_debug_set_stamps(stamps.data_ptr())
ops.sensor_stage(*args)
torch.cuda.synchronize()
lib.rd_debug_set_stamps(None)
s = stamps.cpu().view(8, 16)
names = ["start->embed done", "barrier", "mma1", "epi1", "barrier", "mma2", "epi2", "barrier", "scatter"]
for w in range(2):
    d = [int(s[w, i + 1] - s[w, i]) for i in range(9)]
    print("fwd wg%d" % w, " ".join("%s=%d" % (n, x) for n, x in zip(names, d)), "total", int(s[w, 9] - s[w, 0]))
    print("      embed phase: issue loads=%d masks=%d zero_lds=%d barrier=%d consume=%d" % (
        int(s[w, 10] - s[w, 0]), int(s[w, 11] - s[w, 10]), int(s[w, 12] - s[w, 11]), int(s[w, 13] - s[w, 12]), int(s[w, 1] - s[w, 13])))
# backward: same stamp buffer, overwritten by the backward launch
z = ops.sensor_stage(*args)
zz = z[0] if isinstance(z, (tuple, list)) else z
for p_ in m.parameters(): p_.grad = None
stamps.zero_()
lib.rd_debug_set_stamps(stamps.data_ptr())
zz.backward(torch.randn_like(zz))
torch.cuda.synchronize()
lib.rd_debug_set_stamps(None)
s = stamps.cpu().view(8, 16)
names = ["loads+gather+gate", "barrier", "St->D planes+zero E", "barrier+mma1", "panel issue+epi1", "barrier+dz1save+mma2", "stage dX",
         "barrier+dR_u pass1", "barrier+pass2"]
for w in range(2):
    d = [int(s[w, i + 1] - s[w, i]) for i in range(9)]
    print("bwd wg%d" % w, " ".join("%s=%d" % (n, x) for n, x in zip(names, d)), "total", int(s[w, 9] - s[w, 0]))
    print("      first phase: issue loads=%d zero_lds=%d barrier=%d gather consume=%d gate consume=%d" % (
        int(s[w, 10] - s[w, 0]), int(s[w, 11] - s[w, 10]), int(s[w, 12] - s[w, 11]), int(s[w, 13] - s[w, 12]), int(s[w, 1] - s[w, 13])))
