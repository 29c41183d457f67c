#!/bin/bash
# Run the reference's UNMODIFIED code/Raindrop.py on the GPU with the HIP model behind the `models_rd` shim
# (raindrop_amd/compat_runner.py).  The two reference files are staged -- byte-identical copies -- under _ab/reference_stage/code/
# (git-ignored, shipped by gpurun, never committed):   mkdir -p _ab/reference_stage/code && cp /root/reference/code/{Raindrop.py,utils_rd.py} _ab/reference_stage/code/
# usage (on the GPU box): tools/run_unmodified_script_gpu.sh <outdir under gpurun_out> [P19 samples] [PAM samples]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
ref=$GRAFT_REPO_ROOT/_ab/reference_stage
sha256sum $ref/code/Raindrop.py $ref/code/utils_rd.py > $out/reference_files.sha256
cd $GRAFT_REPO_ROOT
[ "${2:-2400}" != "0" ] && ( time python -m raindrop_amd.compat_runner --dataset P19 --samples ${2:-2400} --root /tmp/ws_p19 --reference $ref ) > $out/P19_script.log 2>&1
echo "P19 rc=$?" >> $out/P19_script.log
( time python -m raindrop_amd.compat_runner --dataset PAM --samples ${3:-640} --root /tmp/ws_pam --reference $ref ) > $out/PAM_script.log 2>&1
echo "PAM rc=$?" >> $out/PAM_script.log
tail -5 $out/P19_script.log; tail -5 $out/PAM_script.log
