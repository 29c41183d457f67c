"""Debug: phase stamps of the slab weight-gradient kernels INSIDE a real training step (operands produced
by the preceding kernels, activations of the whole step competing for the caches), per configuration."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, dp, synth
from raindrop_amd.models_rd import Raindrop_v2
from raindrop_amd.step import TrainStep
lib = _lib.load()
lib.rd_debug_set_wgrad_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
cfg = synth.make_config("P19"); gs = synth.make_structure(cfg, "ones"); B = 256
b = synth.make_batch(cfg, B, seed=1)
dv = {k: (None if v is None else v.to(dev)) for k, v in b.items()}
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], cfg["nhead"], cfg["nhid"], cfg["nlayers"], cfg["dropout"],
                cfg["max_len"], cfg["d_static"], cfg["MAX"], 0.5, cfg["aggreg"], cfg["n_classes"], gs,
                sensor_wise_mask=False).to(dev).train()
live = synth.live_parameter_names(cfg); named = dict(m.named_parameters())
flat = dp.FlatGradAllReduce([(n, named[n]) for n in live])
step = TrainStep(m, flat, dv, use_graph=False)
for _ in range(10): step.run()
torch.cuda.synchronize()
stamps = torch.zeros(5 * 128, dtype=torch.int64, device=dev)
lib.rd_debug_set_wgrad_stamps(stamps.data_ptr()); step.run(); torch.cuda.synchronize(); lib.rd_debug_set_wgrad_stamps(None)
st = stamps.cpu().view(5, 8, 16)
names = ["load+convert s0", "barrier", "mma s0 (+issue s1)", "barrier", "rest of slabs", "epilogue"]
for c in range(5):
    if int(st[c, 0, 0]) == 0: continue
    for w in range(2):
        print("cfg%d wg%d " % (c, w) + " ".join("%s=%d" % (n, int(st[c, w, i + 1] - st[c, w, i])) for i, n in enumerate(names)),
              "total", int(st[c, w, 6] - st[c, w, 0]))
step.close()
