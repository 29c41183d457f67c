#!/bin/bash
# phase-boundary probe (graph kernel boundary vs stream vs persistent kernel + grid barrier) and the GPU suite incl. the float64 tie
d=c2; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 120 tools/_build/probe_boundary 32 256 > $out/probe_boundary.txt 2>&1; echo "probe rc $?" >> $out/probe_boundary.txt
cat $out/probe_boundary.txt
timeout 600 python -m pytest tests/test_token_plan_gpu.py -x -q -s -k float64 > $out/pytest_f64.log 2>&1; echo "f64 rc $?" >> $out/pytest_f64.log
grep -E "float64 tie|passed|failed|rc" $out/pytest_f64.log
