#!/bin/bash
# round 5: variants of the K1 kernels against the default library: stamps in the step + kernel trace, one call
# usage: tools/gpu_k1_variants.sh <outdir> <variant> [<variant> ...]
d=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out; cd $GRAFT_REPO_ROOT
python tools/box_kind.py > $out/box.txt 2>&1; grep -h BOX $out/box.txt
python tools/k1_debug.py 2>&1 | grep -v amdgpu | tail -6
timeout 120 python tools/k1_stamps.py --step 2>&1 | grep -v "amdgpu\|workgroup 1" > $out/stamps_default.txt
for v in "$@"; do
  RD_LIB_PATH=raindrop_amd/_ab/lib_$v.so python tools/k1_debug.py 2>&1 | grep -v amdgpu | tail -6
  RD_LIB_PATH=raindrop_amd/_ab/lib_$v.so timeout 120 python tools/k1_stamps.py --step 2>&1 | grep -v "amdgpu\|workgroup 1" > $out/stamps_$v.txt
done
for rep in 1 2; do
  tools/ktrace.sh gpurun_out/$d/trace_default_$rep.txt 14 -- tools/step_only.py 100
  for v in "$@"; do tools/ktrace.sh gpurun_out/$d/trace_${v}_$rep.txt 14 RD_LIB_PATH=raindrop_amd/_ab/lib_$v.so -- tools/step_only.py 100; done
done
for v in default "$@"; do echo "=== stamps $v"; grep -v "all 256\|start skew" $out/stamps_$v.txt; done
for v in default "$@"; do for rep in 1 2; do echo "== ${v}_$rep"; grep -E "k_msg|k_dw|TOTAL" $out/trace_${v}_$rep.txt | cut -c1-50,90-150; done; done
