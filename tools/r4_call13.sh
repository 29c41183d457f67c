#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/c15; mkdir -p $out; rm -f $out/ab.txt
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/rc.txt
tail -4 $out/pytest.log
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
run() { env "$@" timeout 120 python tools/step_only.py 300 2>&1 | tail -1 | sed "s/^/$* /" >> $out/ab.txt; }
for rep in 1 2 3; do run RD_TRAILING_RIDE=1; run RD_TRAILING_RIDE=0; done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 rocprofv3 --kernel-trace -d $out/kt -o step -- python $R/tools/step_only.py 200 > $out/kt.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) 45 > $out/step_kernel_stats.txt; head -18 $out/step_kernel_stats.txt
rm -rf $out/kt
