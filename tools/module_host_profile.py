"""cProfile of the unchanged loop body on the captured module step (host side): where the ~0.15 ms over TrainStep goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import synth
from raindrop_amd.models_rd import Raindrop_v2
dev = torch.device("cuda")
cfg = synth.make_config("P19")
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, 256, seed=100).items()}
torch.manual_seed(1)
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                synth.make_structure(cfg, "ones")).to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4, **({"fused": True} if os.environ.get("FUSED") == "1" else {}))
crit = torch.nn.CrossEntropyLoss()
def one():
    opt.zero_grad()
    out, _, _ = m.forward(b["src"], b["static"], b["times"], b["lengths"])
    loss = crit(out, b["y"]); loss.backward(); opt.step()
for _ in range(20): one()
torch.cuda.synchronize()
# host-only time per step (no sync inside): enqueue cost
t0 = time.perf_counter()
for _ in range(300): one()
th = time.perf_counter(); torch.cuda.synchronize(); t1 = time.perf_counter()
print("host us/step %.1f   wall ms/step %.4f" % ((th - t0) / 300 * 1e6, (t1 - t0) / 300 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): one()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
