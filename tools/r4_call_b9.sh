#!/bin/bash
# where SYN256's step goes: kernel trace of the bench step (B = 16), + the bench line with the new module-graph field at P19
d=b9; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $out/kt -o step -- python $GRAFT_REPO_ROOT/bench.py --config SYN256 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $out/bench_SYN256.json 2> $out/bench_SYN256.err
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $out/kt -name "*.db" | head -1) 30 > $out/syn256_kernel_stats.txt 2>&1
rm -rf $out/kt
cut -c1-90,95-170 $out/syn256_kernel_stats.txt | head -32
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 50 --warmup 10 > $out/bench_P19.json 2> $out/bench_P19.err
python - <<PY
import json
d=json.loads(open("$out/bench_P19.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["config"]["eager_ms_per_step"], d["config"]["module_graph_ms_per_step"], d["roofline"]["frac"])
PY
