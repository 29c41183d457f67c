#!/bin/bash
# round 5: full GPU suite + step kernel trace of the current library against the round-4 library (base), alternating, one call
# usage: tools/gpu_suite_and_trace.sh <outdir under gpurun_out> [variant libs to trace besides base and default ...]
d=${1:-c1}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out; cd $GRAFT_REPO_ROOT
python tools/box_kind.py > $out/box.txt 2>&1
timeout 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
tail -3 $out/pytest_gpu.txt
for rep in 1 2; do
  tools/ktrace.sh gpurun_out/$d/trace_base_$rep.txt 14 RD_LIB_PATH=raindrop_amd/_ab/lib_base.so -- tools/step_only.py 100
  tools/ktrace.sh gpurun_out/$d/trace_new_$rep.txt 14 -- tools/step_only.py 100
  for v in "$@"; do tools/ktrace.sh gpurun_out/$d/trace_${v}_$rep.txt 14 RD_LIB_PATH=raindrop_amd/_ab/lib_$v.so -- tools/step_only.py 100; done
done
grep -h BOX $out/box.txt
for f in base_1 new_1 base_2 new_2; do echo "== $f"; grep -E "rd::|TOTAL" $out/trace_$f.txt | cut -c1-50,90-150; done
for v in "$@"; do for rep in 1 2; do echo "== ${v}_$rep"; grep -E "rd::|TOTAL" $out/trace_${v}_$rep.txt | cut -c1-50,90-150; done; done
for rep in 1 2 3; do
echo "step base: $(RD_LIB_PATH=raindrop_amd/_ab/lib_base.so timeout 200 python tools/step_only.py 300 2>&1 | tail -1)"
echo "step new: $(timeout 200 python tools/step_only.py 300 2>&1 | tail -1)"
done
