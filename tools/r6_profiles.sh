#!/bin/bash
# Round-6 profile set in ONE gpurun call (~2.5 min of box time): box kind; K1-only kernel trace + FETCH / WRITE / SQ passes; the captured
# step's kernel trace + FETCH / WRITE / SQ passes (separate rocprofv3 runs per counter group, --kernel-trace only); the three JSON files
# bench.py reads (raindrop_amd/k1_pmc_traffic.json, enc_pmc_traffic.json, k1_rocprof.json -- stamped with the sha1 of the kernel
# sources: run this AFTER the last kernel change); the P19 bench line.    usage: tools/r6_profiles.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
R=$GRAFT_REPO_ROOT
python $R/tools/box_kind.py 2>&1 | grep BOX > $out/box.txt
export RD_RG_ROWS32=15 RD_RG_WAVES16=12            # no capture-time tuning: one kernel variant per role in the traces
cd /tmp && export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
k1() { timeout 90 rocprofv3 --kernel-trace ${2:+--pmc $2} -d $out/k1_$1 -o k1 -- python $R/tools/k1_only.py 10 > $out/k1_$1.log 2>&1; }
st() { RD_FULL=1 timeout 150 rocprofv3 --kernel-trace ${2:+--pmc $2} -d $out/st_$1 -o step -- python $R/tools/step_only.py ${3:-8} > $out/st_$1.log 2>&1; }
db() { find $out/$1 -name "*.db" | head -1; }
k1 kt; python $R/tools/rocpd_stats.py $(db k1_kt) 12 > $out/k1_kernel_stats.txt 2>&1
k1 f FETCH_SIZE; k1 w WRITE_SIZE; k1 s "$SQ"
python $R/tools/rocpd_pmc.py $(db k1_f) "rd::" > $out/k1_pmc_fetch.txt 2>&1
python $R/tools/rocpd_pmc.py $(db k1_w) "rd::" > $out/k1_pmc_write.txt 2>&1
python $R/tools/rocpd_pmc.py $(db k1_s) "rd::" > $out/k1_pmc_sq.txt 2>&1
python $R/tools/k1_traffic_json.py $(db k1_f) $(db k1_w) > $out/k1_pmc_traffic.json 2> $out/k1_traffic.err
st kt "" 200; python $R/tools/rocpd_stats.py $(db st_kt) 45 > $out/step_kernel_stats.txt 2>&1
python $R/tools/k1_rocprof_json.py $(db st_kt) $out/box.txt > $out/k1_rocprof.json 2> $out/k1_rocprof.err
st f FETCH_SIZE; st w WRITE_SIZE; st s "$SQ"
python $R/tools/rocpd_pmc.py $(db st_f) "rd::" > $out/step_pmc_fetch.txt 2>&1
python $R/tools/rocpd_pmc.py $(db st_w) "rd::" > $out/step_pmc_write.txt 2>&1
python $R/tools/rocpd_pmc.py $(db st_s) "rd::" > $out/step_pmc_sq.txt 2>&1
python $R/tools/enc_traffic_json.py $(db st_f) $(db st_w) > $out/enc_pmc_traffic.json 2> $out/enc_traffic.err
rm -rf $out/k1_kt $out/k1_f $out/k1_w $out/k1_s $out/st_kt $out/st_f $out/st_w $out/st_s     # the databases are large: keep the summaries
cd $R
unset RD_RG_ROWS32 RD_RG_WAVES16
cp $out/k1_pmc_traffic.json raindrop_amd/k1_pmc_traffic.json
cp $out/enc_pmc_traffic.json raindrop_amd/enc_pmc_traffic.json
cp $out/k1_rocprof.json raindrop_amd/k1_rocprof.json
timeout 400 python bench.py --steps 50 --warmup 10 > $out/bench_P19.json 2> $out/bench_P19.err
python - <<PY
import json
d=json.loads(open("$out/bench_P19.json").read().strip().splitlines()[-1])
r=d["roofline"]; e=d.get("roofline_encoder_layer") or {}
print("P19", d["ms_per_step"], d["value"], "K1 frac", r.get("frac"), "corrected", r.get("frac_boundary_corrected"), "rocprof", r.get("frac_rocprof"),
      "iso", r.get("frac_isolated"), "traffic", r.get("traffic"), "| enc us", e.get("us"), "frac", e.get("frac"), "traffic", e.get("traffic"),
      "| loop default", d["config"].get("module_default_ms_per_step"), "ops", d["config"].get("operator_by_operator_ms_per_step"), d["config"].get("box", {}).get("kind"))
PY
cat $out/box.txt; head -14 $out/step_kernel_stats.txt | cut -c1-60,90-150
