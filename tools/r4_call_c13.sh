#!/bin/bash
d=${1:-c13}; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 120 tools/_build/probe_clocks 2>&1 | grep -E "straight|read stream" | tee $out/box.txt
for t in 1 0 1 0; do
  echo "== RD_CODE_TOUCH=$t" | tee -a $out/head_touch.txt
  RD_CODE_TOUCH=$t timeout 100 python tools/head_warm.py 2>&1 | grep -E "in the step|launch 2" | tee -a $out/head_touch.txt
done
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "head" 2>&1 | tail -2 | tee $out/pytest_head.txt
