#!/bin/bash
# PMC passes over the hipGraph training step (tools/step_only.py): FETCH_SIZE, WRITE_SIZE, SQ busy / MFMA busy / LDS conflicts.
# usage: tools/step_pmc.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { timeout 90 rocprofv3 --kernel-trace --pmc $2 -d $out/$1 -o step -- python $GRAFT_REPO_ROOT/tools/step_only.py 8 > $out/$1.log 2>&1;
        python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $out/$1 -name "*.db" | head -1) "rd::" > $out/pmc_$1.txt 2>&1; }
run f "FETCH_SIZE"
run w "WRITE_SIZE"
run s "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
