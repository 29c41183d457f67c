#!/bin/bash
# round 5, call a2: K1 phase stamps with all 16 waves + every workgroup's start / end (isolated and in the step); what the bare launch costs
out=$GRAFT_REPO_ROOT/gpurun_out/a2; mkdir -p $out; cd $GRAFT_REPO_ROOT
python tools/box_kind.py > $out/box.txt 2>&1
timeout 120 python tools/k1_stamps.py > $out/k1_stamps_isolated.txt 2>&1
timeout 120 python tools/k1_stamps.py --step > $out/k1_stamps_step.txt 2>&1
tools/ktrace.sh gpurun_out/a2/k1_trace_default.txt 6 -- tools/k1_only.py 10
tools/ktrace.sh gpurun_out/a2/k1_trace_empty.txt 6 RD_LIB_PATH=raindrop_amd/_ab/lib_k1empty.so -- tools/k1_only.py 10
tools/ktrace.sh gpurun_out/a2/step_trace_default.txt 16 -- tools/step_only.py 100
grep -h -v amdgpu.ids $out/box.txt $out/k1_stamps_isolated.txt $out/k1_stamps_step.txt $out/k1_trace_default.txt $out/k1_trace_empty.txt $out/step_trace_default.txt
