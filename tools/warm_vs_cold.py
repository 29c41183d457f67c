"""How much of each kernel of the step is COLD INSTRUCTION FETCH?  The eager step with every C-ABI call issued TWICE in a row: the
second issue finds its kernels' code where the first left it (instruction cache / L2), the first finds what a whole step of other
kernels left.  Run under `rocprofv3 --kernel-trace`, then `python tools/warm_vs_cold.py --report <db>` prints, per kernel, the mean
duration of the first and of the second dispatch of each pair.   RD_TRAILING_RIDE=0 (riders would only exist in the first issue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def report(path):
    import sqlite3
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    rows = c.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, disp, sym)).fetchall()
    # the call pattern is c(); c(): a call that launches kernels K1..Kn gives K1..Kn K1..Kn -- per kernel name the dispatches
    # alternate first / second issue
    per = {}
    for name, s, e in rows:
        per.setdefault(name, []).append((e - s) / 1e3)
    print("%-70s %6s %10s %10s %8s" % ("kernel", "pairs", "first_us", "second_us", "ratio"))
    tot1 = tot2 = 0.0
    for name, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        if not name.startswith("rd::") and "rd::" not in name:
            continue
        d = d[len(d) // 2 // 2 * 2:]                      # second half of the run (steady state), an even count
        a, b = d[0::2], d[1::2]
        n = min(len(a), len(b))
        if n == 0:
            continue
        m1, m2 = sum(a[:n]) / n, sum(b[:n]) / n
        tot1 += m1 * n; tot2 += m2 * n
        print("%-70s %6d %10.2f %10.2f %8.2f" % (name.replace("(anonymous namespace)::", "")[:70], n, m1, m2, m1 / m2))
    print("sum over the listed kernels: first %.1f us, second %.1f us" % (tot1, tot2))


if len(sys.argv) > 2 and sys.argv[1] == "--report":
    report(sys.argv[2]); sys.exit(0)

import torch
from raindrop_amd import dp, synth
from raindrop_amd.models_rd import Raindrop_v2
from raindrop_amd.step import TrainStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
cfg = synth.make_config("P19")
torch.manual_seed(1)
m = Raindrop_v2(cfg["d_inp"], cfg["d_model"], 2, cfg["nhid"], 2, 0.2, cfg["max_len"], cfg["d_static"], 100, 0.5, "mean", 2,
                synth.make_structure(cfg, "ones")).to(dev).train()
b = {k: (None if v is None else v.to(dev)) for k, v in synth.make_batch(cfg, 256, seed=100).items()}
named = dict(m.named_parameters())
flat = dp.FlatGradAllReduce([(n, named[n]) for n in synth.live_parameter_names(cfg)], n_buckets=2)
ts = TrainStep(m, flat, b, use_graph=False, autotune=False)
orig = ts._call
FLUSH = os.environ.get("WVC_FLUSH") == "1"      # a 1-GB fill between the two issues: data caches cold again, the instruction cache not
big = torch.empty(1 << 28, dtype=torch.float32, device=dev) if FLUSH else None
def twice(name, *a):
    orig(name, *a)
    if FLUSH:
        big.fill_(1.0)
    orig(name, *a)
ts._call = twice
for _ in range(steps):
    ts.run()
torch.cuda.synchronize()
print("done", float(ts.loss))
