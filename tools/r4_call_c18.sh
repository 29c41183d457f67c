#!/bin/bash
d=${1:-c18}; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
tools/_build/probe_clocks 2>&1 | grep -E "straight" | head -1 | tee $out/box.txt
timeout 300 python bench.py --steps 50 --warmup 10 > $out/bench_P19.json 2> $out/bench_P19.err
python - <<PY
import json
d=json.loads(open("$out/bench_P19.json").read().strip().splitlines()[-1])
r=d["roofline"]; e=d.get("roofline_encoder_layer") or {}
print("P19", d["ms_per_step"], d["value"], "K1 frac", r.get("frac"), "us", r.get("us"), "traffic", r.get("traffic"), "enc us", e.get("us"), "enc frac_live", e.get("frac_live_rows"))
PY
