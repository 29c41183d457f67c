#!/bin/bash
# round 4, third session: the trimmed profile set for the tree with the own-code touch (K1-only FETCH / WRITE passes, graph-step kernel
# trace + FETCH / WRITE passes, traffic JSONs) and the P19 bench line.  usage: tools/r4_final2.sh <outdir under gpurun_out>
d=$1; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
R=$GRAFT_REPO_ROOT
$R/tools/_build/probe_clocks 2>&1 | grep -E "straight|stream" > $out/box.txt
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
cd /tmp && export TMPDIR=/tmp
k1() { timeout 90 rocprofv3 --kernel-trace ${2:+--pmc $2} -d $out/k1_$1 -o k1 -- python $R/tools/k1_only.py 10 > $out/k1_$1.log 2>&1; }
st() { timeout 120 rocprofv3 --kernel-trace ${2:+--pmc $2} -d $out/st_$1 -o step -- python $R/tools/step_only.py ${3:-8} > $out/st_$1.log 2>&1; }
db() { find $out/$1 -name "*.db" | head -1; }
k1 f FETCH_SIZE; k1 w WRITE_SIZE
python $R/tools/rocpd_stats.py $(db k1_f) 12 > $out/k1_kernel_stats.txt 2>&1
python $R/tools/rocpd_pmc.py $(db k1_f) "rd::" > $out/k1_pmc_fetch.txt 2>&1
python $R/tools/rocpd_pmc.py $(db k1_w) "rd::" > $out/k1_pmc_write.txt 2>&1
python $R/tools/k1_traffic_json.py $(db k1_f) $(db k1_w) > $out/k1_pmc_traffic.json 2> $out/k1_traffic.err
st kt "" 200; python $R/tools/rocpd_stats.py $(db st_kt) 45 > $out/step_kernel_stats.txt 2>&1
st f FETCH_SIZE; st w WRITE_SIZE
python $R/tools/rocpd_pmc.py $(db st_f) "rd::" > $out/step_pmc_fetch.txt 2>&1
python $R/tools/rocpd_pmc.py $(db st_w) "rd::" > $out/step_pmc_write.txt 2>&1
python $R/tools/enc_traffic_json.py $(db st_f) $(db st_w) > $out/enc_pmc_traffic.json 2> $out/enc_traffic.err
rm -rf $out/k1_f $out/k1_w $out/st_kt $out/st_f $out/st_w
cd $R
unset RD_RG_ROWS32 RD_RG_WAVES16
cp $out/k1_pmc_traffic.json raindrop_amd/k1_pmc_traffic.json
cp $out/enc_pmc_traffic.json raindrop_amd/enc_pmc_traffic.json
timeout 300 python bench.py --steps 50 --warmup 10 > $out/bench_P19.json 2> $out/bench_P19.err
python - <<PY
import json
d=json.loads(open("$out/bench_P19.json").read().strip().splitlines()[-1])
r=d["roofline"]; e=d.get("roofline_encoder_layer") or {}
print("P19", d["ms_per_step"], d["value"], "K1 frac", r.get("frac"), "traffic", r.get("traffic"), "enc us", e.get("us"), "enc frac_live", e.get("frac_live_rows"), "enc traffic", e.get("traffic"))
PY
cat $out/box.txt; head -16 $out/step_kernel_stats.txt
