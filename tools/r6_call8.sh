cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; out=gpurun_out/c8; mkdir -p $out
( time RD_BENCH_KEEP_TRACE=$out timeout 1200 python bench.py > $out/bench.json 2> $out/bench.err ) 2>&1 | tail -3
tail -c 400 $out/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/c8/bench.json") if x.startswith("{")]
d=json.loads(l[-1]); r=d["roofline"]
print(d["ms_per_step"], d["value"], r.get("frac"), r.get("frac_source","")[:60], r.get("frac_events"))
print(json.dumps(d["config"].get("other_configs"), indent=1))
print(d["config"].get("module_graph_ms_per_step"), d["config"].get("host_us_per_step"), d["config"]["box"]["kind"], d["config"].get("cpu_baseline_kind","")[:80])
PY
