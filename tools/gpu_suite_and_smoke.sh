cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; out=gpurun_out/f1; mkdir -p $out
python tools/box_kind.py 2>&1 | grep BOX | cut -c1-160
( time timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1 ) 2>&1 | grep real; tail -3 $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
