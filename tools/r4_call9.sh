#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/c9; mkdir -p $out; rm -f $out/ab.txt
cd $GRAFT_REPO_ROOT
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
run() { env "$@" timeout 120 python tools/step_only.py 300 2>&1 | tail -1 | sed "s/^/$* /" >> $out/ab.txt; }
for rep in 1 2 3; do run RD_ATTN_FUSE=1; run RD_ATTN_FUSE=0; run RD_ATTN_FUSE=0 RD_ENC_LEAN=0; done
cat $out/ab.txt
