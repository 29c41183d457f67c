#!/bin/bash
# head kernel: request order (rows before W0), 192 x 192 fetch, interleaved wave sums -- tests, phase stamps, kernel-trace A/B against the library of HEAD
d=${1:-c6}; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_token_plan_gpu.py tests/test_graph_module_gpu.py -x -q -k "head or golden or float64 or static_train_step or replayed or module" > $out/pytest_head.log 2>&1; echo "rc $?" >> $out/pytest_head.log
tail -4 $out/pytest_head.log
tools/_build/probe_clocks 2>&1 | grep -E "shader|stream" | tee $out/box.txt
timeout 200 python tools/head_timing.py 2>&1 | grep -v amdgpu.ids | tee $out/head_stamps_new.txt
RD_LIB_PATH=raindrop_amd/_ab/lib_oldhead.so timeout 200 python tools/head_timing.py 2>&1 | grep -v amdgpu.ids | tee $out/head_stamps_old.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
for v in new old new old; do
  lib=""; [ $v = old ] && lib=$R/raindrop_amd/_ab/lib_oldhead.so
  RD_LIB_PATH=$lib timeout 120 rocprofv3 --kernel-trace -d $out/kt_$v -o step -- python $R/tools/step_only.py 200 > $out/kt_$v.log 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_$v -name "*.db" | head -1) 14 > $out/kt_$v.txt 2>&1
  echo "== $v: $(tail -1 $out/kt_$v.log)"; grep -E "k_head_rows|TOTAL" $out/kt_$v.txt
  rm -rf $out/kt_$v
done
