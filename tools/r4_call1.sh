#!/bin/bash
# round 4, call 1: full GPU suite on the new parity tests, P19 bench (in-step K1 roofline), P12 bf16x3 vs bf16, step kernel trace
out=$GRAFT_REPO_ROOT/gpurun_out/c1; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 -x -k "token_plan or dp_gpu" > $out/pytest_new.log 2>&1; echo "pytest_new rc=$?" >> $out/rc.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 > $out/pytest_all.log 2>&1; echo "pytest_all rc=$?" >> $out/rc.txt
timeout 600 python bench.py --steps 50 --warmup 10 > $out/bench_P19.json 2> $out/bench_P19.err; echo "bench rc=$?" >> $out/rc.txt
for pr in bf16x3 bf16 bf16x3 bf16; do
  timeout 300 python bench.py --config P12 --batch 256 --precision $pr --steps 30 --warmup 5 --no-cpu-baseline --no-roofline >> $out/bench_P12_$pr.json 2>> $out/bench_P12.err
done
timeout 300 python bench.py --feed --steps 30 --warmup 5 --no-cpu-baseline --no-roofline > $out/bench_feed.json 2> $out/bench_feed.err; echo "feed rc=$?" >> $out/rc.txt
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 rocprofv3 --kernel-trace -d $out/st_kt -o step -- python $R/tools/step_only.py 200 > $out/st_kt.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/st_kt -name "*.db" | head -1) 45 > $out/step_kernel_stats.txt 2>&1
rm -rf $out/st_kt
tail -3 $out/pytest_new.log; tail -3 $out/pytest_all.log; cat $out/rc.txt
