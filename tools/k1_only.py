"""Run only the K1 path (sensor stage forward + backward, P19 B=256, on the step's token plan exactly as bench.py's roofline loop
runs it) a few times: target for rocprofv3 kernel-trace / PMC passes (HBM FETCH_SIZE / WRITE_SIZE per launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raindrop_amd import _lib, ops, synth
from raindrop_amd.models_rd import Raindrop_v2
dev = torch.device("cuda")
cfg = synth.make_config("P19"); B = 256
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                synth.make_structure(cfg, "ones")).to(dev)
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, B, seed=100).items()}
g = m._graph(dev); shp = _lib.shape(B, 60, 34, 4)
det = [t.detach() for t in (b["src"], b["times"], b["lengths"], m.pos_encoder.timescales(dev), g["ssum"], m.R_u,
                            m.ob_propagation.lin_value.weight, m.ob_propagation.lin_value.bias,
                            m.ob_propagation_layer2.lin_value.weight, m.ob_propagation_layer2.lin_value.bias)]
dz = torch.randn(60, B, 152, device=dev)
plan = bench._make_plan(shp, b["lengths"]) if bench._use_token_plan(cfg) else None
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    with bench._plan_scope(plan):
        z, _, saved = ops.sensor_stage_fwd_raw(*det, shp, 0.2, 1234)
        ops.sensor_stage_bwd_raw(det[0], det[5], det[6], det[8], det[4], saved, z, dz, shp, 0.2)
torch.cuda.synchronize()
