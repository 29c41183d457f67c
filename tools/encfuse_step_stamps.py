"""Debug: phase stamps (clock64, all 16 waves of workgroup 0) of the fused row-local encoder kernels (rd_encfuse.hip) as they run INSIDE
the captured P19 training step (token plan, LEAN chains, tall blocks): the last launch of each kind wins (forward: layer 1, backward:
layer 0).  Usage: encfuse_step_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, dp, synth
from raindrop_amd.models_rd import Raindrop_v2
from raindrop_amd.optim import FlatAdam
from raindrop_amd.step import TrainStep
lib = _lib.load()
lib.rd_debug_set_encfuse_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
cfg = synth.make_config("P19")
torch.manual_seed(1)
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                synth.make_structure(cfg, "ones")).to(dev).train()
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, 256, seed=100).items()}
named = dict(m.named_parameters())
flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
opt = FlatAdam(flat.flatten_parameters(), lr=1e-4)
stamps = torch.zeros(8192, dtype=torch.int64, device=dev)


def run(which):
    """one TrainStep captured with the stamps of ONE kernel kind live (both kinds share the buffer)"""
    lib.rd_debug_set_encfuse_stamps(stamps.data_ptr())
    ts = TrainStep(m, flat, b, autotune=False)
    lib.rd_debug_set_encfuse_stamps(None)
    for _ in range(4):
        ts.run(); opt.step()
    torch.cuda.synchronize()
    stamps.zero_()
    ts.run(); opt.step()
    torch.cuda.synchronize()
    return stamps.cpu()


def show(tag, s, n, off=0):
    ph = s[off:off + 256].view(16, 16)
    t0 = int(ph[:, 0].min())
    print(tag, "-- cycles since the first wave started: min .. max over the 16 waves")
    for i in range(n):
        col = ph[:, i] - t0
        print("  %2d  %7d .. %7d" % (i, int(col.min()), int(col.max())))


s = run("both")
show("in-step k_enc_post_fwd (layer 1): 0 start | 1 rows split | 2 barrier | 3 out_proj staged | 4 barrier | 5 LN1 done | 6 barrier | "
     "7 linear1 + h epilogue done | 8 barrier | 9 linear2 staged | 10 barrier | 11 LN2 done   (before the accumulator epilogues: 7 linear1 "
     "staged | 8 barrier | 9 h done | 10 barrier | 11 linear2 staged | 12 barrier | 13 LN2 done)", s, 12)
show("in-step k_enc_pre_bwd (layer 0): 0 start | 1 barrier | 2 LN2' done | 3 barrier | 4 du product + gate done | 5 barrier | 6 dx1 staged | "
     "7 barrier | 8 LN1' done | 9 barrier | 10 d attn stored   (before: 4 du staged | 5 barrier | 6 gate done | 7 barrier | 8 dx1 staged | "
     "9 barrier | 10 LN1' done | 11 barrier | 12 d attn staged | 13 barrier | 14 stored)", s, 11, 4096)
