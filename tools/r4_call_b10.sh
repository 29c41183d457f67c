#!/bin/bash
# split-K weight-gradient tile size at the big shapes
d=b10; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
for t in 64 128; do
  echo "SYN256 tile=$t $(RD_WGRAD_TILE=$t python bench.py --config SYN256 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
  echo "P12    tile=$t $(RD_WGRAD_TILE=$t python bench.py --config P12 --batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
  echo "PAM    tile=$t $(RD_WGRAD_TILE=$t python bench.py --config PAM --batch 64 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"])')"
done 2>&1 | tee $out/ab.log
