#!/bin/bash
# model-level use_beta beyond 64 sensors (golden wide80_beta_sparse) + split-launch width A/B
d=b4; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "use_beta" > $out/pytest_new.log 2>&1; echo "new rc $?" >> $out/pytest_new.log
tail -5 $out/pytest_new.log
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
for i in 1 2; do
  for gx in 64 32 16 8; do echo "gx=$gx $(RD_WSPLIT_GX=$gx python tools/step_only.py 300 2>/dev/null | tail -1)"; done
done > $out/ab.log 2>&1
cat $out/ab.log
cd /tmp && export TMPDIR=/tmp
for gx in 64 32 16 8; do
  RD_WSPLIT_GX=$gx timeout 120 rocprofv3 --kernel-trace -d $out/kt$gx -o step -- python $GRAFT_REPO_ROOT/tools/step_only.py 100 > $out/kt$gx.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $out/kt$gx -name "*.db" | head -1) 16 > $out/step_kernel_stats_gx$gx.txt 2>&1
  rm -rf $out/kt$gx
  grep -h "k_wsplit" $out/step_kernel_stats_gx$gx.txt | cut -c1-50,95-170
done
