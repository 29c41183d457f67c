#!/bin/bash
# first-launch experiment: plan workgroup last (0) / first (1) / both (2), alternating, + the beta tests again
d=b3; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_graph_beta_gpu.py tests/test_token_plan_gpu.py -x -q > $out/pytest_new.log 2>&1; echo "new rc $?" >> $out/pytest_new.log
tail -5 $out/pytest_new.log
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
for i in 1 2 3; do
  for pf in 0 1; do echo "plan_first=$pf $(RD_PLAN_FIRST=$pf python tools/step_only.py 300 2>/dev/null | tail -1)"; done
done > $out/ab.log 2>&1
cat $out/ab.log
cd /tmp && export TMPDIR=/tmp
for pf in 0 1; do
  RD_PLAN_FIRST=$pf timeout 120 rocprofv3 --kernel-trace -d $out/kt$pf -o step -- python $GRAFT_REPO_ROOT/tools/step_only.py 100 > $out/kt$pf.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $out/kt$pf -name "*.db" | head -1) 16 > $out/step_kernel_stats_pf$pf.txt 2>&1
  rm -rf $out/kt$pf
done
unset RD_RG_ROWS32 RD_RG_WAVES16
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_all.log 2>&1; echo "all rc $?" >> $out/pytest_all.log
tail -4 $out/pytest_all.log
