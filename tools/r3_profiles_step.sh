#!/bin/bash
# The graph-step half of tools/r3_profiles.sh (kernel trace + FETCH / WRITE / SQ passes over tools/step_only.py, encoder traffic JSON):
# enough when only encoder-side kernel sources changed (the K1 traffic file is stamped with the K1 sources only).
# usage: tools/r3_profiles_step.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
export RD_RG_ROWS32=15 RD_RG_WAVES16=12            # no capture-time tuning: one kernel variant per role in the traces
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
st() { timeout 120 rocprofv3 --kernel-trace ${2:+--pmc $2} -d $out/st_$1 -o step -- python $R/tools/step_only.py ${3:-8} > $out/st_$1.log 2>&1; }
db() { find $out/$1 -name "*.db" | head -1; }
st kt "" 200; python $R/tools/rocpd_stats.py $(db st_kt) 45 > $out/step_kernel_stats.txt 2>&1
st f FETCH_SIZE; st w WRITE_SIZE; st s "$SQ"
python $R/tools/rocpd_pmc.py $(db st_f) "rd::" > $out/step_pmc_fetch.txt 2>&1
python $R/tools/rocpd_pmc.py $(db st_w) "rd::" > $out/step_pmc_write.txt 2>&1
python $R/tools/rocpd_pmc.py $(db st_s) "rd::" > $out/step_pmc_sq.txt 2>&1
python $R/tools/enc_traffic_json.py $(db st_f) $(db st_w) > $out/enc_pmc_traffic.json 2> $out/enc_traffic.err
rm -rf $out/st_kt $out/st_f $out/st_w $out/st_s
