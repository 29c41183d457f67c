#!/bin/bash
# round 5: the fused encoder chains (rd_encfuse.hip) of the default library against _ab/lib_base.so and any named variants, one call:
# chain parity tests, the captured step's loss, phase stamps of both chains INSIDE the step, kernel traces (alternating), step times.
# usage: tools/gpu_enc_variants.sh <outdir under gpurun_out> [variant ...]
d=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out; cd $GRAFT_REPO_ROOT
python tools/box_kind.py > $out/box.txt 2>&1; grep -h BOX $out/box.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_token_plan_gpu.py -x -q -k "fused_row_local_chains or encoder_tile_weight or encoder_layer or benchmarked_step or static_train_step" > $out/pytest_enc.txt 2>&1
tail -3 $out/pytest_enc.txt
libs=(base default "$@")
lp() { if [ "$1" = default ]; then echo ""; else echo "RD_LIB_PATH=raindrop_amd/_ab/lib_$1.so"; fi; }
# QUICK=1: stamps of the LAST library only, one trace and two step times per library
for v in "${libs[@]}"; do
  if [ -n "$QUICK" ] && [ $v != "${libs[-1]}" ]; then echo "(skipped)" > $out/stamps_$v.txt; continue; fi
  env $(lp $v) timeout 120 python tools/encfuse_step_stamps.py 2>&1 | grep -v amdgpu > $out/stamps_$v.txt
done
for rep in 1 ${QUICK:+} $([ -z "$QUICK" ] && echo 2); do
  for v in "${libs[@]}"; do tools/ktrace.sh gpurun_out/$d/trace_${v}_$rep.txt 14 $(lp $v) -- tools/step_only.py 100; done
done
for v in "${libs[@]}"; do echo "=== stamps $v"; cat $out/stamps_$v.txt; done
for v in "${libs[@]}"; do for rep in 1 2; do [ -f $out/trace_${v}_$rep.txt ] || continue; echo "== ${v}_$rep"; grep -E "rd::|TOTAL" $out/trace_${v}_$rep.txt | cut -c1-50,90-150; done; done
for rep in 1 2 $([ -z "$QUICK" ] && echo 3); do
  for v in "${libs[@]}"; do echo "step $v: $(env $(lp $v) timeout 200 python tools/step_only.py 300 2>&1 | tail -1)"; done
done
