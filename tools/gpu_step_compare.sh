#!/bin/bash
# first-step loss + gradients and the next losses of the captured P19 step: default library against _ab/lib_base.so (tools/step_dump.py)
# usage: tools/gpu_step_compare.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out; cd $GRAFT_REPO_ROOT
RD_LIB_PATH=raindrop_amd/_ab/lib_base.so timeout 200 python tools/step_dump.py $out/base.npy 8 2>&1 | grep -v amdgpu | tail -2
timeout 200 python tools/step_dump.py $out/new.npy 8 2>&1 | grep -v amdgpu | tail -2
python tools/step_dump.py --compare $out/base.npy $out/new.npy
rm -f $out/base.npy $out/new.npy
