#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/c4; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_token_plan_gpu.py -m gpu -q --maxfail=8 > $out/pytest_plan.log 2>&1; echo "pytest_plan rc=$?" >> $out/rc.txt
tail -40 $out/pytest_plan.log
for pr in bf16x3 bf16; do for tp in 1 0; do
  RD_TOKEN_PLAN=$tp timeout 300 python bench.py --config P12 --batch 256 --precision $pr --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>>$out/bench_P12.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('P12 $pr plan=$tp', d['ms_per_step'], d['value'])" >> $out/p12.txt
done; done
cat $out/p12.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 > $out/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> $out/rc.txt
tail -8 $out/pytest_all.log; cat $out/rc.txt
