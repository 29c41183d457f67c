#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/c10; mkdir -p $out; rm -f $out/ab.txt
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_token_plan_gpu.py tests/test_dp_gpu.py -m gpu -q --maxfail=8 > $out/pytest_plan.log 2>&1; echo "pytest rc=$?" >> $out/rc.txt
tail -5 $out/pytest_plan.log
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
run() { env "$@" timeout 120 python tools/step_only.py 300 2>&1 | tail -1 | sed "s/^/$* /" >> $out/ab.txt; }
for rep in 1 2 3; do run RD_SIDE_REDUCE=1; run RD_SIDE_REDUCE=0; done
cat $out/ab.txt
