#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/c3; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_token_plan_gpu.py -m gpu -q -x > $out/pytest_plan.log 2>&1; echo "pytest_plan rc=$?" >> $out/rc.txt
tail -5 $out/pytest_plan.log
python tools/attnfuse_timing.py > $out/stamps.txt 2>&1; cat $out/stamps.txt | tail -52
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
for v in 1 0 1 0; do RD_ATTN_FUSE=$v timeout 120 python tools/step_only.py 300 2>&1 | tail -1 | sed "s/^/fuse=$v /" >> $out/ab.txt; done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 rocprofv3 --kernel-trace -d $out/st_kt -o step -- python $R/tools/step_only.py 200 > $out/st_kt.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/st_kt -name "*.db" | head -1) 45 > $out/step_kernel_stats.txt 2>&1
rm -rf $out/st_kt
head -18 $out/step_kernel_stats.txt; cat $out/rc.txt
