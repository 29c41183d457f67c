#!/bin/bash
# unfused K1 weight gradients on the tile stream: tests, P12 / SYN256 A/B
d=b14; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_graph_module_gpu.py -x -q -k "streamed or wide_encoder or golden or recaptures" > $out/pytest_new.log 2>&1; echo "new rc $?" >> $out/pytest_new.log
tail -5 $out/pytest_new.log
for t in 1 0 1 0; do
  echo "P12 bf16x3 stream=$t $(RD_K1_WGRAD_STREAM=$t python bench.py --config P12 --batch 256 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
  echo "P12 bf16   stream=$t $(RD_K1_WGRAD_STREAM=$t python bench.py --config P12 --batch 256 --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done 2>&1 | tee $out/ab.log
for t in 1 0; do
  echo "SYN256 stream=$t $(RD_K1_WGRAD_STREAM=$t python bench.py --config SYN256 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done 2>&1 | tee -a $out/ab.log
