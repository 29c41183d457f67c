#!/bin/bash
# the whole GPU suite on the final tree, then the final profile set + bench lines
d=b15; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_all.log 2>&1; echo "all rc $?" >> $out/pytest_all.log
tail -4 $out/pytest_all.log
bash tools/r4_final.sh f6
