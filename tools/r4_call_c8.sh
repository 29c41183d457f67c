#!/bin/bash
d=${1:-c8}; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 120 tools/_build/probe_clocks 2>&1 | tee $out/box.txt
timeout 100 python tools/step_only.py 400 2>&1 | tail -1 | tee -a $out/box.txt
timeout 100 python tools/head_timing.py 2>&1 | grep -v amdgpu.ids | tee $out/head_stamps.txt
