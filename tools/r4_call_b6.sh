#!/bin/bash
# phase stamps of the head rows kernel and of the fused attention kernels, current tree
d=b6; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
python tools/head_timing.py > $out/head_stamps.txt 2>&1
python tools/attnfuse_timing.py > $out/attnfuse_stamps.txt 2>&1
cat $out/head_stamps.txt | tail -14
