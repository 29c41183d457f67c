#!/bin/bash
d=c4; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower --showclocks --showperflevel > $out/smi_idle.txt 2>&1
timeout 200 python tools/clock_ramp.py 30 1000 2>&1 | tee $out/clock_ramp.txt
rocm-smi -a 2>/dev/null | grep -iE "clock|power|perf|level|cap|freq" | head -60 > $out/smi_all.txt
