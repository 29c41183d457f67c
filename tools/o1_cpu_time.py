"""Time the reference's own files (oracle O1: /root/reference under the PyG shim) on this host: forward + CE + backward,
train mode, dropout 0.2, one batch -- the CPU figure SURVEY 8d asks for.  Build container only (needs /root/reference).
    python tools/o1_cpu_time.py P19 256  >> merges into raindrop_amd/o1_cpu_baseline.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from oracle import ref_loader
from raindrop_amd import synth
name, B = sys.argv[1], int(sys.argv[2])
torch.set_num_threads(os.cpu_count())
cfg = synth.make_config(name)
model = ref_loader.build_raindrop_v2(cfg, synth.make_structure(cfg, "ones").clone())
model.train()
b = synth.make_batch(cfg, B, seed=100)
ts = []
for i in range(4):
    t0 = time.perf_counter()
    for p in model.parameters():
        p.grad = None
    logits, _, _ = ref_loader.forward(model, b["src"], b["static"], b["times"], b["lengths"])
    F.cross_entropy(logits, b["y"]).backward()
    ts.append(time.perf_counter() - t0)
med = sorted(ts[1:])[1]
path = os.path.join(ROOT, "raindrop_amd", "o1_cpu_baseline.json")
d = json.load(open(path)) if os.path.exists(path) else {}
d["%s_B%d" % (name, B)] = {"value": round(B / med, 1), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "reference",
                           "where": "build container (no GPU): %d logical CPUs" % os.cpu_count(),
                           "sample": "%d %s-shaped samples, one batch, fwd+CE+bwd, train mode dropout 0.2, median of 3 after 1 warm-up; "
                                     "the reference's own code/models_rd.py + Ob_propagation.py under oracle/pyg_shim" % (B, name)}
json.dump(d, open(path, "w"), indent=1, sort_keys=True)
print(d["%s_B%d" % (name, B)])
