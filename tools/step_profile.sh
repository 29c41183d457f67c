#!/bin/bash
# kernel trace of the hipGraph training step: tools/step_profile.sh <outdir under gpurun_out> [steps]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
n=${2:-20}
cd /tmp && export TMPDIR=/tmp
timeout 90 rocprofv3 --kernel-trace -d $out/kt -o step -- python $GRAFT_REPO_ROOT/tools/step_only.py $n > $out/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) 45 > $out/kernels.txt 2>&1
cut -c1-86,90-150 $out/kernels.txt | head -45
