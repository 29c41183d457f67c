out=$GRAFT_REPO_ROOT/gpurun_out/r3h; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 90 rocprofv3 --kernel-trace -d $out/kt -o ef -- python $GRAFT_REPO_ROOT/tools/encfuse_timing.py > $out/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) 12 > $out/kernels.txt 2>&1
cut -c1-100,108-150 $out/kernels.txt | head -8
head -20 $out/kt.log
