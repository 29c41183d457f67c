#!/bin/bash
# the UNMODIFIED code/Raindrop.py on the GPU, eager module path vs RD_MODULE_GRAPH=1; then the GPU suite
d=b8; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
ref=$GRAFT_REPO_ROOT/_ab/reference_stage
sha256sum $ref/code/Raindrop.py $ref/code/utils_rd.py > $out/reference_files.sha256
( time python -m raindrop_amd.compat_runner --dataset P19 --samples 3000 --root /tmp/ws_p19 --reference $ref ) > $out/P19_script_eager.log 2>&1
echo "rc=$?" >> $out/P19_script_eager.log
( time RD_MODULE_GRAPH=1 python -m raindrop_amd.compat_runner --dataset P19 --samples 3000 --root /tmp/ws_p19g --reference $ref ) > $out/P19_script_module_graph.log 2>&1
echo "rc=$?" >> $out/P19_script_module_graph.log
grep -h "Total Time\|Testing: AUROC\|^real\|rc=" $out/P19_script_eager.log $out/P19_script_module_graph.log
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_all.log 2>&1; echo "all rc $?" >> $out/pytest_all.log
tail -4 $out/pytest_all.log
