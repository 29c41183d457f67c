#!/bin/bash
# round 5, call a1: K1 phase stamps of the current tree; RD_TOUCH_ALL A/B for the configurations outside the P19 step
out=$GRAFT_REPO_ROOT/gpurun_out/a1; mkdir -p $out; cd $GRAFT_REPO_ROOT
python tools/box_kind.py > $out/box.txt 2>&1
timeout 120 python tools/fused_timing.py > $out/k1_stamps.txt 2>&1
TA=raindrop_amd/_ab/lib_touchall.so
for rep in 1 2; do
  echo "P12 bf16 default: $(RD_PRECISION=bf16 timeout 200 python tools/step_cfg.py P12 256 60 2>&1 | tail -1)" >> $out/touchall_ab.txt
  echo "P12 bf16 touchall: $(RD_LIB_PATH=$TA RD_PRECISION=bf16 timeout 200 python tools/step_cfg.py P12 256 60 2>&1 | tail -1)" >> $out/touchall_ab.txt
done
for rep in 1 2; do
  echo "PAM default: $(timeout 200 python tools/step_cfg.py PAM 64 30 2>&1 | tail -1)" >> $out/touchall_ab.txt
  echo "PAM touchall: $(RD_LIB_PATH=$TA timeout 200 python tools/step_cfg.py PAM 64 30 2>&1 | tail -1)" >> $out/touchall_ab.txt
done
echo "SYN256 default: $(timeout 200 python tools/step_cfg.py SYN256 16 10 2>&1 | tail -1)" >> $out/touchall_ab.txt
echo "SYN256 touchall: $(RD_LIB_PATH=$TA timeout 200 python tools/step_cfg.py SYN256 16 10 2>&1 | tail -1)" >> $out/touchall_ab.txt
echo "P19 default: $(timeout 200 python tools/step_only.py 300 2>&1 | tail -1)" >> $out/touchall_ab.txt
echo "P19 touchall: $(RD_LIB_PATH=$TA timeout 200 python tools/step_only.py 300 2>&1 | tail -1)" >> $out/touchall_ab.txt
cat $out/box.txt $out/k1_stamps.txt $out/touchall_ab.txt
