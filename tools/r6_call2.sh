cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; out=gpurun_out/c2; mkdir -p $out
python tools/box_kind.py > $out/box.txt 2>&1
timeout 600 python -m pytest tests/test_trajectory_gpu.py -x -q -s > $out/traj.txt 2>&1; tail -25 $out/traj.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or adam" > $out/golden.txt 2>&1; tail -5 $out/golden.txt
timeout 600 python tools/grad_gate_diag.py > $out/gate_diag.txt 2>&1; grep -c . $out/gate_diag.txt
for r in 1 2; do RD_FULL=1 timeout 200 python tools/step_only.py 300 2>&1 | tail -1; done
tools/ktrace.sh gpurun_out/c2/trace_full.txt 16 RD_FULL=1 -- tools/step_only.py 100
grep -E "rd::|TOTAL" $out/trace_full.txt | cut -c1-50,90-150
