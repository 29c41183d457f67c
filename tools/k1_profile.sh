#!/bin/bash
# K1-only profile on the GPU box: kernel trace + FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes) over tools/k1_only.py.  usage: tools/k1_profile.sh <outdir under gpurun_out> [pmc]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 90 rocprofv3 --kernel-trace -d $out/kt -o k1 -- python $GRAFT_REPO_ROOT/tools/k1_only.py 20 > $out/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) 12 > $out/kernels.txt 2>&1
if [ "$2" = "pmc" ]; then
  timeout 90 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pf -o k1 -- python $GRAFT_REPO_ROOT/tools/k1_only.py 10 > $out/pf.log 2>&1
  timeout 90 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pw -o k1 -- python $GRAFT_REPO_ROOT/tools/k1_only.py 10 > $out/pw.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $out/pf -name "*.db" | head -1) "rd::" > $out/pmc_fetch.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $out/pw -name "*.db" | head -1) "rd::" > $out/pmc_write.txt 2>&1
fi
cut -c1-70,90-150 $out/kernels.txt | head -10
