#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/c8; mkdir -p $out
cd $GRAFT_REPO_ROOT
for v in "RD_ATTN_FUSE=1 RD_ENC_LEAN=1" "RD_ATTN_FUSE=0 RD_ENC_LEAN=1" "RD_ATTN_FUSE=1 RD_ENC_LEAN=0" "RD_ATTN_FUSE=0 RD_ENC_LEAN=0" "RD_ATTN_FUSE=1 RD_ENC_LEAN=1"; do
  env $v timeout 300 python bench.py --k1-child --batch 256 --config P19 2>/dev/null | grep ENCROOFLINE | python -c "import sys,json; d=json.loads(sys.stdin.read().split(' ',1)[1]); print('$v', 'layer us', d['us'])"
done
