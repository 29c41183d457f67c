#!/bin/bash
# round 5, call b4: K1 with 8 waves (two column tiles per wave) against 16 waves, in one call
out=$GRAFT_REPO_ROOT/gpurun_out/b4; mkdir -p $out; cd $GRAFT_REPO_ROOT
python tools/box_kind.py > $out/box.txt 2>&1
RD_LIB_PATH=raindrop_amd/_ab/lib_nthr512.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_token_plan_gpu.py -x -q -m gpu -k "k1 or sensor_stage or model_vs_golden or benchmarked_step" > $out/pytest_k1.txt 2>&1
tail -2 $out/pytest_k1.txt
RD_LIB_PATH=raindrop_amd/_ab/lib_nthr512.so timeout 120 python tools/k1_stamps.py --step > $out/k1_stamps_step_512.txt 2>&1
for rep in 1 2; do
  tools/ktrace.sh gpurun_out/b4/trace_new_$rep.txt 12 -- tools/step_only.py 100
  tools/ktrace.sh gpurun_out/b4/trace_512_$rep.txt 12 RD_LIB_PATH=raindrop_amd/_ab/lib_nthr512.so -- tools/step_only.py 100
done
grep -h -v amdgpu.ids $out/box.txt $out/k1_stamps_step_512.txt | grep -v "workgroup 1" 
for f in new_1 512_1 new_2 512_2; do echo "== $f"; grep -E "k_msg|TOTAL" $out/trace_$f.txt | cut -c1-60,90-150; done
