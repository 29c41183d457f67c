# usage: prof_exp.sh <tag> ; env passed through; prints the kernel table top lines matching a pattern
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 90 rocprofv3 --kernel-trace -d $out/kt -o step -- python $GRAFT_REPO_ROOT/tools/step_only.py 20 > $out/kt.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) 45 > $out/kernels.txt 2>&1
echo "== $1"; cut -c1-60,108-150 $out/kernels.txt | grep -i "${2:-enc_}"
