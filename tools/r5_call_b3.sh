#!/bin/bash
# round 5, call b3: K1 A/B inside one call -- round-4 library (base) | new kernels with group B's panel before (blate0) / behind (default) the first barrier
out=$GRAFT_REPO_ROOT/gpurun_out/b3; mkdir -p $out; cd $GRAFT_REPO_ROOT
python tools/box_kind.py > $out/box.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_token_plan_gpu.py -x -q -m gpu -k "k1 or sensor_stage or model_vs_golden or benchmarked_step or token_plan or edge_cases" > $out/pytest_k1.txt 2>&1
tail -2 $out/pytest_k1.txt
timeout 120 python tools/k1_stamps.py > $out/k1_stamps_isolated.txt 2>&1
timeout 120 python tools/k1_stamps.py --step > $out/k1_stamps_step.txt 2>&1
for rep in 1 2; do
  tools/ktrace.sh gpurun_out/b3/trace_base_$rep.txt 12 RD_LIB_PATH=raindrop_amd/_ab/lib_base.so -- tools/step_only.py 100
  tools/ktrace.sh gpurun_out/b3/trace_blate0_$rep.txt 12 RD_LIB_PATH=raindrop_amd/_ab/lib_blate0.so -- tools/step_only.py 100
  tools/ktrace.sh gpurun_out/b3/trace_new_$rep.txt 12 -- tools/step_only.py 100
done
grep -h -v amdgpu.ids $out/box.txt $out/k1_stamps_isolated.txt $out/k1_stamps_step.txt | grep -v "workgroup 1" 
for f in base_1 blate0_1 new_1 base_2 blate0_2 new_2; do echo "== $f"; grep -E "k_msg|k_dw|TOTAL" $out/trace_$f.txt | cut -c1-60,90-150; done
for rep in 1 2; do
echo "step base: $(RD_LIB_PATH=raindrop_amd/_ab/lib_base.so timeout 200 python tools/step_only.py 300 2>&1 | tail -1)"
echo "step new: $(timeout 200 python tools/step_only.py 300 2>&1 | tail -1)"
done
