#!/bin/bash
# per-kernel A/B of the own-code touch (kernel trace of the graph step), alternating; then a bench line
d=${1:-c17}; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
R=$GRAFT_REPO_ROOT
$R/tools/_build/probe_clocks 2>&1 | grep -E "straight" | head -1 | tee $out/box.txt
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
cd /tmp && export TMPDIR=/tmp
for v in touch notouch touch notouch; do
  lib=""; [ $v = notouch ] && lib=$R/raindrop_amd/_ab/lib_notouch.so
  RD_LIB_PATH=$lib timeout 120 rocprofv3 --kernel-trace -d $out/kt_$v -o step -- python $R/tools/step_only.py 200 > $out/kt_$v.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_$v -name "*.db" | head -1) 13 > $out/kt_$v.txt 2>&1
  echo "== $v $(grep ms/step $out/kt_$v.log)" | tee -a $out/ab.txt
  awk 'NR>1 && NR<14 {n=$0; sub(/^.*::/,"",n); print substr($0,1,0) $(NF-3), $(NF-2), substr(n,1,40)}' $out/kt_$v.txt | tee -a $out/ab.txt
  rm -rf $out/kt_$v
done
