#!/bin/bash
# what kind of box: clocks / latencies / stream rates, partition modes; then the head's phase stamps and a short step timing on the same box
d=${1:-c5}; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
{ timeout 120 tools/_build/probe_clocks; rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -i partition; } 2>&1 | tee $out/box.txt
timeout 200 python tools/head_timing.py 2>&1 | grep -v amdgpu.ids | tee $out/head_stamps.txt
timeout 200 python tools/step_only.py 400 2>&1 | tail -1 | tee $out/step.txt
