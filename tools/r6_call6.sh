cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; out=gpurun_out/c6; mkdir -p $out
python tools/box_kind.py > $out/box.txt 2>&1; grep -h BOX $out/box.txt | cut -c1-200
timeout 900 python -m pytest tests/test_graph_module_gpu.py tests/test_trajectory_gpu.py tests/test_train_loop_gpu.py tests/test_dp_gpu.py -x -q > $out/pytest.txt 2>&1; tail -4 $out/pytest.txt
for v in 0 1 0 1; do echo "== RD_MODULE_GRAD_VIEWS=$v"; RD_MODULE_GRAD_VIEWS=$v timeout 300 python tools/module_step_timing.py 2>&1 | grep "module graph"; done
echo "== fused torch Adam"; RD_TIMING_FUSED_ADAM=1 timeout 300 python tools/module_step_timing.py 2>&1 | grep "ms/step"
timeout 300 python tools/host_side_timing.py 300 2>&1 | tail -3
