#!/bin/bash
# the captured step behind the module surface: tests, then its cost next to the eager path and to TrainStep
d=b7; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_graph_module_gpu.py -x -q > $out/pytest_new.log 2>&1; echo "new rc $?" >> $out/pytest_new.log
tail -30 $out/pytest_new.log
timeout 300 python tools/module_step_timing.py > $out/module_timing.log 2>&1
tail -8 $out/module_timing.log
