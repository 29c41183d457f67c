#!/bin/bash
# kernel-trace A/B (new head vs the library of HEAD), alternating; prints k_head_rows avg / min and the step's kernel total
d=${1:-c7}; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
$R/tools/_build/probe_clocks 2>&1 | grep -E "shader|stream" | tee $out/box.txt
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|fclk|Power" | tee -a $out/box.txt
for v in new old new old new old; do
  lib=""; [ $v = old ] && lib=$R/raindrop_amd/_ab/lib_oldhead.so
  RD_LIB_PATH=$lib timeout 120 rocprofv3 --kernel-trace -d $out/kt_$v -o step -- python $R/tools/step_only.py 300 > $out/kt_$v.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_$v -name "*.db" | head -1) 14 > $out/kt_$v.txt 2>&1
  echo "== $v: $(grep -E 'k_head_rows' $out/kt_$v.txt | awk '{print "head avg", $(NF-3), "min", $(NF-2)}') $(grep TOTAL $out/kt_$v.txt)" | tee -a $out/ab.txt
  rm -rf $out/kt_$v
done
