"""The reference's loop body (model.forward, cross entropy, loss.backward(), torch.optim.Adam.step) on the P19 benchmark batch:
eager operator-by-operator autograd against the captured module step (raindrop_amd/graph_module.py).   [B] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import synth
from raindrop_amd.models_rd import Raindrop_v2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda")
cfg = synth.make_config("P19")
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, B, seed=100).items()}
for graph in (False, True, True):
    torch.manual_seed(1)
    m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                    synth.make_structure(cfg, "ones")).to(dev).train()
    m.graph_step = graph
    fused = os.environ.get("RD_TIMING_FUSED_ADAM", "0") == "1"      # what raindrop_amd.compat_runner selects for an unmodified script
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, **({"fused": True} if fused else {}))
    crit = torch.nn.CrossEntropyLoss()

    def one(with_opt=True):
        opt.zero_grad()
        out, _, _ = m.forward(b["src"], b["static"], b["times"], b["lengths"])
        loss = crit(out, b["y"])
        loss.backward()
        if with_opt:
            opt.step()
        return loss
    for mode in ("fwd+loss+bwd", "fwd+loss+bwd+torch Adam"):
        for _ in range(5):
            one(mode.endswith("Adam"))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            l = one(mode.endswith("Adam"))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print("%-14s %-26s %.4f ms/step  (loss %.5f)" % ("module graph" if graph else "eager autograd", mode, (t1 - t0) * 1e3 / steps, float(l)))
