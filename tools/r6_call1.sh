cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; out=gpurun_out/c1; mkdir -p $out
python tools/box_kind.py > $out/box.txt 2>&1
timeout 1200 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -5 $out/pytest_gpu.txt
RD_BENCH_KEEP_TRACE=$out timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
tail -c 600 $out/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/c1/bench.json") if x.startswith("{")]
d=json.loads(l[-1]); r=d["roofline"]
print(d["ms_per_step"], d["value"], r.get("frac"), r.get("frac_source"), r.get("frac_events"), r.get("frac_stamped"))
print(json.dumps(r.get("rocprof"))[:1500])
print(d["config"].get("module_graph_ms_per_step"), d["config"].get("host_us_per_step"), d["config"]["box"]["kind"])
PY
cat $out/bench_step_kernel_stats.txt | cut -c1-60,92-140 | head -20
