#!/bin/bash
# final tree: touch A/B (step), whole GPU suite
d=${1:-c16}; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 120 tools/_build/probe_clocks 2>&1 | grep -E "straight|read stream" | tee $out/box.txt
for v in touch notouch touch notouch; do
  lib=""; [ $v = notouch ] && lib=$GRAFT_REPO_ROOT/raindrop_amd/_ab/lib_notouch.so
  echo "$v $(RD_LIB_PATH=$lib timeout 100 python tools/step_only.py 600 2>&1 | tail -1)" | tee -a $out/ab.txt
done
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest_all.log 2>&1; echo "all rc $?" >> $out/pytest_all.log
tail -3 $out/pytest_all.log
