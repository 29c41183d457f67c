cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; out=gpurun_out/c3; mkdir -p $out
timeout 600 python -m pytest tests/test_trajectory_gpu.py -q -s > $out/traj.txt 2>&1; grep -v "Warning\|warn\|^$\|detach" $out/traj.txt | tail -40
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q > $out/parity.txt 2>&1; tail -5 $out/parity.txt
