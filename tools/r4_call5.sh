#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/c5; mkdir -p $out
cd $GRAFT_REPO_ROOT
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
run() { env "$@" timeout 120 python tools/step_only.py 300 2>&1 | tail -1 | sed "s/^/$* /" >> $out/ab.txt; }
for rep in 1 2; do
run RD_X=0
run RD_HEAD_RB1=0
run RD_WSPLIT_GX=16
run RD_WSPLIT_GX=32
run RD_WSPLIT_GX=8
done
cat $out/ab.txt
