#!/bin/bash
d=${1:-c12}; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export RD_RG_ROWS32=15 RD_RG_WAVES16=12 RD_TRAILING_RIDE=0
$R/tools/_build/probe_clocks 2>&1 | grep -E "L2 hit\)|straight|read stream" | head -4 | tee $out/box.txt
for f in 0 1; do
WVC_FLUSH=$f timeout 120 rocprofv3 --kernel-trace -d $out/kt$f -o step -- python $R/tools/warm_vs_cold.py 20 > $out/kt$f.log 2>&1
echo "== flush between the issues: $f" | tee -a $out/warm_vs_cold.txt
python $R/tools/warm_vs_cold.py --report $(find $out/kt$f -name "*.db" | head -1) 2>&1 | tee -a $out/warm_vs_cold.txt
rm -rf $out/kt$f
done
timeout 100 python $R/tools/step_only.py 400 2>&1 | tail -1 | tee -a $out/box.txt
