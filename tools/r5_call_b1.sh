#!/bin/bash
# round 5, call b1 / b2 (same script; b2 = groups rebalanced): K1 with swizzled planes / plane-sourced row tiles / group B's panel behind the barrier: parity subset, stamps, kernel trace
out=$GRAFT_REPO_ROOT/gpurun_out/b2; mkdir -p $out; cd $GRAFT_REPO_ROOT
python tools/box_kind.py > $out/box.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_token_plan_gpu.py -x -q -m gpu -k "k1 or sensor_stage or model_vs_golden or pe_and_mask or benchmarked_step or token_plan or edge_cases or static_train_step" > $out/pytest_k1.txt 2>&1
tail -5 $out/pytest_k1.txt
timeout 120 python tools/k1_stamps.py --step > $out/k1_stamps_step.txt 2>&1
tools/ktrace.sh gpurun_out/b2/step_trace.txt 16 -- tools/step_only.py 100
grep -h -v amdgpu.ids $out/box.txt $out/k1_stamps_step.txt $out/step_trace.txt
echo "step: $(timeout 200 python tools/step_only.py 300 2>&1 | tail -1)"
