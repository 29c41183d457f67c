"""Debug: phase stamps of the fused message-passing kernels (rd_msgpass_fused.hip RD_STAMP), all 16 waves of the four longest samples'
workgroups, and the start / end of EVERY workgroup (dispatch skew, tail).
    python tools/k1_stamps.py            # the K1 launches alone, on the step's token plan (buffers re-used: cache-warm)
    python tools/k1_stamps.py --step     # inside the captured training step (cold operands, cold code)
"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from raindrop_amd import _lib, dp, ops, synth
from raindrop_amd.models_rd import Raindrop_v2
lib = _lib.load()
dev = torch.device("cuda")
cfg = synth.make_config("P19"); B = 256
W = 4 * 16 * 16 + 4 * 1024
stamps = torch.zeros(2 * W, dtype=torch.int64, device=dev)
lib.rd_debug_set_stamps.argtypes = [ctypes.c_void_p]
torch.manual_seed(1)
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                synth.make_structure(cfg, "ones")).to(dev).train()
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, B, seed=100).items()}

FWD = [(0, "start (own code touched)"), (10, "loads issued, pads zeroed"), (11, "masks made, W1 panel issued (A)"), (1, "observations consumed"),
       (2, "barrier 1"), (13, "[B: W1 panel issued, PE + mask written]"), (3, "GEMM1 (+ W2 first half issued)"), (12, "W2 second half issued"),
       (14, "[A: X row tiles stored]"), (4, "epilogue 1"), (5, "barrier 2"), (6, "Y1 row tiles + GEMM2"), (7, "epilogue 2"), (8, "barrier 3"),
       (9, "scatter z + gate bytes (end)")]
BWD = [(0, "start"), (10, "gather + W2^T panel (A) issued, gate masks loaded"), (11, "pads zeroed"), (13, "dz consumed -> D planes"), (1, "barrier 1"),
       (2, "[B: W2^T panel issued]"), (4, "GEMM dZ2 W2 (+ W1^T first half issued)"), (14, "W1^T 2nd half, dR_u loads [A: dZ2 row tiles]"),
       (5, "epilogue dZ1"), (15, "barrier 2"), (6, "dZ1 row tiles + GEMM dZ1 W1"), (7, "dX -> staging"), (8, "barrier 3 + dR_u pass 1"),
       (9, "barrier 4 + pass 2 (end)")]


def show(tag, s, names):
    ph = s[:1024].view(4, 16, 16)
    wg = s[1024:].view(1024, 4)[:B]
    for w in range(2):
        t0 = int(ph[w, :, 0].min())
        print("%s workgroup %d: cycles since its first wave's start stamp; min..max over group A (waves 0-7) | group B (waves 8-15)" % (tag, w))
        for idx, nm in names:
            col = ph[w, :, idx] - t0
            print("   %2d %-38s %6d .. %6d  | %6d .. %6d" % (idx, nm, int(col[:8].min()), int(col[:8].max()), int(col[8:].min()), int(col[8:].max())))
    dur = (wg[:, 2] - wg[:, 0]).float()                        # cycles, per workgroup
    st = (wg[:, 1] - wg[:, 1].min()).float() * 10.0            # ns (100 MHz wall clock)
    en = (wg[:, 3] - wg[:, 1].min()).float() * 10.0
    q = lambda x, p: float(x.kthvalue(max(1, int(round(p * (len(x) - 1))) + 1)).values)
    print("%s all %d workgroups: in-kernel cycles min %.0f  median %.0f  max %.0f  (rank 0 %.0f, rank %d %.0f)" % (
        tag, B, dur.min(), q(dur, 0.5), dur.max(), dur[0], B - 1, dur[B - 1]))
    print("%s start skew (ns after the first start): median %.0f  p90 %.0f  max %.0f;   end (ns after the first start): min %.0f median %.0f max %.0f" % (
        tag, q(st, 0.5), q(st, 0.9), st.max(), en.min(), q(en, 0.5), en.max()))


if "--step" in sys.argv:
    from raindrop_amd.optim import FlatAdam
    from raindrop_amd.step import TrainStep
    named = dict(m.named_parameters())
    flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
    opt = FlatAdam(flat.flatten_parameters(), lr=1e-4)
    lib.rd_debug_set_stamps(stamps.data_ptr())                 # read at enqueue = at capture
    ts = TrainStep(m, flat, b, autotune=False)
    lib.rd_debug_set_stamps(None)
    for _ in range(5):
        ts.run(); opt.step()
    torch.cuda.synchronize()
    stamps.zero_()                                             # the workgroup END stamps are atomicMax'es and clock64 is per XCD: keep ONE replay's
    ts.run(); opt.step()
    torch.cuda.synchronize()
    tag = "in-step"
else:
    g = m._graph(dev); shp = _lib.shape(B, 60, 34, 4)
    det = [t.detach() for t in (b["src"], b["times"], b["lengths"], m.pos_encoder.timescales(dev), g["ssum"], m.R_u,
                                m.ob_propagation.lin_value.weight, m.ob_propagation.lin_value.bias,
                                m.ob_propagation_layer2.lin_value.weight, m.ob_propagation_layer2.lin_value.bias)]
    dz = torch.randn(60, B, 152, device=dev)
    plan = bench._make_plan(shp, b["lengths"]) if bench._use_token_plan(cfg) else None
    for it in range(4):
        if it == 3:
            lib.rd_debug_set_stamps(stamps.data_ptr())
        with bench._plan_scope(plan):
            z, _, saved = ops.sensor_stage_fwd_raw(*det, shp, 0.2, 1234)
            ops.sensor_stage_bwd_raw(det[0], det[5], det[6], det[8], det[4], saved, z, dz, shp, 0.2)
    torch.cuda.synchronize()
    lib.rd_debug_set_stamps(None)
    tag = "isolated"
s = stamps.cpu()
show(tag + " fwd", s[:W], FWD)
show(tag + " bwd", s[W:], BWD)
