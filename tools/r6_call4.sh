cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; out=gpurun_out/c4; mkdir -p $out
python tools/box_kind.py > $out/box.txt 2>&1; grep -h BOX $out/box.txt | cut -c1-200
timeout 900 python -m pytest tests/test_token_plan_gpu.py tests/test_gpu_parity.py tests/test_graph_module_gpu.py tests/test_trajectory_gpu.py -x -q > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
for rep in 1 2; do
  tools/ktrace.sh gpurun_out/c4/trace_pre0_$rep.txt 14 RD_FULL=1 RD_STEP_PRE=0 -- tools/step_only.py 100
  tools/ktrace.sh gpurun_out/c4/trace_pre1_$rep.txt 14 RD_FULL=1 -- tools/step_only.py 100
done
for f in pre0_1 pre1_1 pre0_2 pre1_2; do echo "== $f"; grep -E "k_msg|k_wsplit|k_dw|k_attn_fwd|TOTAL" $out/trace_$f.txt | cut -c1-50,90-150; done
for rep in 1 2 3; do
echo "step pre0: $(RD_FULL=1 RD_STEP_PRE=0 timeout 200 python tools/step_only.py 300 2>&1 | tail -1)"
echo "step pre1: $(RD_FULL=1 timeout 200 python tools/step_only.py 300 2>&1 | tail -1)"
done
