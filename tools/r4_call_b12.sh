#!/bin/bash
# kernel traces of the P12 (B = 256, bf16x3) and SYN256 (B = 16) steps
d=b12; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $out/kt -o step -- python $GRAFT_REPO_ROOT/bench.py --config P12 --batch 256 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $out/bench_P12.json 2> $out/bench_P12.err
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) 24 > $out/p12_kernel_stats.txt 2>&1
rm -rf $out/kt
cut -c1-90,95-170 $out/p12_kernel_stats.txt | head -26
timeout 300 rocprofv3 --kernel-trace -d $out/kt2 -o step -- python $GRAFT_REPO_ROOT/bench.py --config SYN256 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $out/bench_SYN256.json 2> $out/bench_SYN256.err
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $out/kt2 -name "*.db" | head -1) 16 > $out/syn256_kernel_stats.txt 2>&1
rm -rf $out/kt2
cut -c1-90,95-170 $out/syn256_kernel_stats.txt | head -18
