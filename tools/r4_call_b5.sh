#!/bin/bash
# riders in the two-graph form: DP + token-plan tests, split-vs-one-graph step time
d=b5; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_token_plan_gpu.py -x -q > $out/pytest_new.log 2>&1; echo "new rc $?" >> $out/pytest_new.log
tail -5 $out/pytest_new.log
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
for i in 1 2; do
  echo "one graph      $(python tools/step_only.py 300 2>/dev/null | tail -1)"
  echo "split, riders  $(RD_SPLIT=1 python tools/step_only.py 300 2>/dev/null | tail -1)"
  echo "split, no ride $(RD_SPLIT=1 RD_TRAILING_RIDE=0 python tools/step_only.py 300 2>/dev/null | tail -1)"
done > $out/ab.log 2>&1
cat $out/ab.log
