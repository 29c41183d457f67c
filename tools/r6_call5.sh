cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; out=gpurun_out/c5; mkdir -p $out
python tools/box_kind.py > $out/box.txt 2>&1; grep -h BOX $out/box.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_token_plan_gpu.py tests/test_dp_gpu.py tests/test_trajectory_gpu.py -x -q > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
for rep in 1 2; do
  tools/ktrace.sh gpurun_out/c5/trace_old_$rep.txt 14 RD_FULL=1 RD_K1_BWD_RIDER=0 RD_FOLD_REDUCE=0 -- tools/step_only.py 100
  tools/ktrace.sh gpurun_out/c5/trace_rider_$rep.txt 14 RD_FULL=1 RD_FOLD_REDUCE=0 -- tools/step_only.py 100
  tools/ktrace.sh gpurun_out/c5/trace_new_$rep.txt 14 RD_FULL=1 -- tools/step_only.py 100
done
for f in old_1 rider_1 new_1 old_2 rider_2 new_2; do echo "== $f"; grep -E "k_msg|k_wsplit|k_dw|k_adam|TOTAL" $out/trace_$f.txt | cut -c1-50,90-150; done
for rep in 1 2 3; do
echo "step old: $(RD_FULL=1 RD_K1_BWD_RIDER=0 RD_FOLD_REDUCE=0 timeout 200 python tools/step_only.py 300 2>&1 | tail -1)"
echo "step rider: $(RD_FULL=1 RD_FOLD_REDUCE=0 timeout 200 python tools/step_only.py 300 2>&1 | tail -1)"
echo "step new: $(RD_FULL=1 RD_STEP_ONLY_PLAIN_ADAM=0 timeout 200 python tools/step_only.py 300 2>&1 | tail -1)"
done
