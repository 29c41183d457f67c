#!/bin/bash
# Round-4 profile set in ONE gpurun call: K1-only kernel trace + FETCH / WRITE / SQ passes, graph-step kernel trace + FETCH / WRITE / SQ
# passes (separate rocprofv3 runs per counter group, --kernel-trace only).  usage: tools/r4_profiles.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
export RD_RG_ROWS32=15 RD_RG_WAVES16=12            # no capture-time tuning: one kernel variant per role in the traces
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
k1() { timeout 90 rocprofv3 --kernel-trace ${2:+--pmc $2} -d $out/k1_$1 -o k1 -- python $R/tools/k1_only.py 10 > $out/k1_$1.log 2>&1; }
st() { timeout 120 rocprofv3 --kernel-trace ${2:+--pmc $2} -d $out/st_$1 -o step -- python $R/tools/step_only.py ${3:-8} > $out/st_$1.log 2>&1; }
db() { find $out/$1 -name "*.db" | head -1; }
k1 kt; python $R/tools/rocpd_stats.py $(db k1_kt) 12 > $out/k1_kernel_stats.txt 2>&1
k1 f FETCH_SIZE; k1 w WRITE_SIZE; k1 s "$SQ"
python $R/tools/rocpd_pmc.py $(db k1_f) "rd::" > $out/k1_pmc_fetch.txt 2>&1
python $R/tools/rocpd_pmc.py $(db k1_w) "rd::" > $out/k1_pmc_write.txt 2>&1
python $R/tools/rocpd_pmc.py $(db k1_s) "rd::" > $out/k1_pmc_sq.txt 2>&1
python $R/tools/k1_traffic_json.py $(db k1_f) $(db k1_w) > $out/k1_pmc_traffic.json 2> $out/k1_traffic.err
st kt "" 200; python $R/tools/rocpd_stats.py $(db st_kt) 45 > $out/step_kernel_stats.txt 2>&1
st f FETCH_SIZE; st w WRITE_SIZE; st s "$SQ"
python $R/tools/rocpd_pmc.py $(db st_f) "rd::" > $out/step_pmc_fetch.txt 2>&1
python $R/tools/rocpd_pmc.py $(db st_w) "rd::" > $out/step_pmc_write.txt 2>&1
python $R/tools/rocpd_pmc.py $(db st_s) "rd::" > $out/step_pmc_sq.txt 2>&1
python $R/tools/enc_traffic_json.py $(db st_f) $(db st_w) > $out/enc_pmc_traffic.json 2> $out/enc_traffic.err
rm -rf $out/k1_kt $out/k1_f $out/k1_w $out/k1_s $out/st_kt $out/st_f $out/st_w $out/st_s     # the databases are large: keep the summaries
ls -la $out | head -30
