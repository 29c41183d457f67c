#!/bin/bash
d=${1:-c11}; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export RD_RG_ROWS32=15 RD_RG_WAVES16=12 RD_TRAILING_RIDE=0
$R/tools/_build/probe_clocks 2>&1 | grep -E "L2 hit\)|straight|read stream" | head -4 | tee $out/box.txt
timeout 120 rocprofv3 --kernel-trace -d $out/kt -o step -- python $R/tools/warm_vs_cold.py 30 > $out/kt.log 2>&1
tail -2 $out/kt.log
python $R/tools/warm_vs_cold.py --report $(find $out/kt -name "*.db" | head -1) 2>&1 | tee $out/warm_vs_cold.txt
rm -rf $out/kt
