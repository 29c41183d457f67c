#!/bin/bash
# whole GPU suite on the final tree, then the default bench line as the driver runs it (timed)
d=c3; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest_all.log 2>&1; echo "all rc $?" >> $out/pytest_all.log
tail -3 $out/pytest_all.log
s=$(date +%s.%N)
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err
e=$(date +%s.%N)
echo "bench wall $(echo "$e - $s" | bc) s" | tee $out/bench_wall.txt
python - <<PY
import json
d=json.loads(open("$out/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], "K1 frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), "enc", (d.get("roofline_encoder_layer") or {}).get("frac_live_rows"), "cpu", d["cpu_baseline"]["value"])
PY
