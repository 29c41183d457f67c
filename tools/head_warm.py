"""Is k_head_rows bound by COLD INSTRUCTION FETCH?  Phase stamps of the kernel as it runs inside the step (its code was last
executed a whole step = ~900 MB of traffic ago) against the same launch repeated back to back (code warm in the instruction cache /
L2).  Usage: head_warm.py [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raindrop_amd import _lib, dp, synth
from raindrop_amd.models_rd import Raindrop_v2
from raindrop_amd.step import TrainStep
lib = _lib.load()
lib.rd_debug_set_head_stamps.argtypes = [ctypes.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
cfg = synth.make_config("P19")
torch.manual_seed(1)
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                synth.make_structure(cfg, "ones")).to(dev).train()
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, B, seed=100).items()}
named = dict(m.named_parameters())
flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
ts = TrainStep(m, flat, b, use_graph=False, autotune=False)
saved = {}
orig = ts._call
def spy(name, *a):
    if name == "rd_head_train":
        saved["args"] = a
    orig(name, *a)
ts._call = spy
for _ in range(3):
    ts.run()
torch.cuda.synchronize()
stamps = torch.zeros(32, dtype=torch.int64, device=dev)
N = ["start", "requests issued", "rows summed", "barrier", "mean + emb", "hid", "logits", "softmax / loss", "dhid", "dfeat partials", "dfeat reduced", "dr rows out"]
def show(tag):
    s = stamps.cpu().tolist()
    print(tag, "total %6d |" % (s[11] - s[0]), " ".join("%s +%d" % (N[i], s[i] - s[i - 1]) for i in range(1, 12)))
lib.rd_debug_set_head_stamps(stamps.data_ptr())
ts.run(); torch.cuda.synchronize(); show("in the step (cold code)      ")
plan = getattr(ts, "plan", None)
from raindrop_amd import ops
if plan is not None:
    _lib.call("rd_set_token_plan", ctypes.c_void_p(plan.data_ptr()))
_lib.call("rd_set_defer_trailing", 0)
for rep in range(4):
    _lib.call("rd_head_train", *saved["args"])
    torch.cuda.synchronize(); show("back to back, launch %d       " % rep)
# data warm, code cold: a 1-GB fill in between evicts code AND data; then twice more
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
big.fill_(1.0); torch.cuda.synchronize()
_lib.call("rd_head_train", *saved["args"]); torch.cuda.synchronize(); show("after a 1-GB fill (all cold)  ")
_lib.call("rd_head_train", *saved["args"]); torch.cuda.synchronize(); show("again (all warm)              ")
lib.rd_debug_set_head_stamps(None)
