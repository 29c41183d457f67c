"""Debug: per-phase cycle stamps of the fused message-passing forward and backward kernels: every wave of the first
4 workgroups; prints, per phase boundary, the time since kernel start as min / max over the waves of each group
(A = waves 0-3, B = waves 4-7) of workgroup 0.  Usage: fused_timing.py [B]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, ops, synth
from raindrop_amd.models_rd import Raindrop_v2
lib = _lib.load()
dev = torch.device("cuda")
cfg = synth.make_config("P19")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gs = synth.make_structure(cfg, "ones")
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2, gs).to(dev)
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, B, seed=0).items()}
g = m._graph(dev)
shp = _lib.shape(B, 60, 34, 4)
stamps = torch.zeros(4 * 8 * 16, dtype=torch.int64, device=dev)
args = (b["src"], b["times"], b["lengths"], m.pos_encoder.timescales(dev), g["ssum"], m.R_u,
        m.ob_propagation.lin_value.weight, m.ob_propagation.lin_value.bias,
        m.ob_propagation_layer2.lin_value.weight, m.ob_propagation_layer2.lin_value.bias, shp, 0.2, 1)
for _ in range(3):
    ops.sensor_stage(*args)
lib.rd_debug_set_stamps.argtypes = [ctypes.c_void_p]


def show(tag, order, names):
    s = stamps.cpu().view(4, 8, 16)
    for wg in range(1):
        t0 = int(s[wg, :, 0].min())
        print("%s wg%d: cycles since the first wave started, min..max over group A (waves 0-3) | group B (waves 4-7)" % (tag, wg))
        for idx, nm in zip(order, names):
            col = s[wg, :, idx] - t0
            print("   %2d %-36s %6d .. %6d  | %6d .. %6d" % (idx, nm, int(col[:4].min()), int(col[:4].max()), int(col[4:].min()), int(col[4:].max())))


lib.rd_debug_set_stamps(stamps.data_ptr())
ops.sensor_stage(*args)
torch.cuda.synchronize()
lib.rd_debug_set_stamps(None)
show("fwd", [0, 10, 11, 1, 2, 13, 3, 12, 14, 4, 5, 6, 7, 8, 9],
     ["start", "B: masks+embed done", "W1 panel issued", "embed phase done", "barrier", "B: X row tiles stored",
      "GEMM1 (+W2 first half issued)", "W2 second half issued", "A: X row tiles stored", "epilogue 1", "barrier", "GEMM2", "epilogue 2", "barrier",
      "scatter z + PE (end)"])
z = ops.sensor_stage(*args)
zz = z[0] if isinstance(z, (tuple, list)) else z
for p_ in m.parameters(): p_.grad = None
stamps.zero_()
lib.rd_debug_set_stamps(stamps.data_ptr())
zz.backward(torch.randn_like(zz))
torch.cuda.synchronize()
lib.rd_debug_set_stamps(None)
show("bwd", [0, 10, 11, 12, 13, 1, 2, 3, 4, 14, 5, 15, 6, 7, 8, 9],
     ["start", "dz gather + W2^T panel issued", "pads zeroed, masks in LDS", "barrier", "dz -> staging", "barrier",
      "staging -> D planes", "barrier", "GEMM (dZ2 W2) [B: after row tiles]", "W1^T 2nd half + dR_u loads issued", "epilogue (dZ1) [A: after row tiles]",
      "barrier", "GEMM (dZ1 W1)", "dX -> staging", "barrier + dR_u pass 1", "barrier + pass 2 (end)"])
