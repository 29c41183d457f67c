#!/bin/bash
# generic-path weight gradients on the tile stream: parity tests, then SYN256 / PAM / P12 with and without
d=b11; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_all.log 2>&1; echo "all rc $?" >> $out/pytest_all.log
tail -6 $out/pytest_all.log
for t in 1 0; do
  echo "SYN256 tw2=$t $(RD_TILE_WGRAD_GENERIC=$t python bench.py --config SYN256 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
  echo "PAM    tw2=$t $(RD_TILE_WGRAD_GENERIC=$t python bench.py --config PAM --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done 2>&1 | tee $out/ab.log
