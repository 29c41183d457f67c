#!/bin/bash
# HIP runtime knobs against the captured step (tools/step_only.py, 400 steps each, two alternations): does any of them change what a
# kernel boundary inside the graph costs?
d=c1e; out=$GRAFT_REPO_ROOT/gpurun_out/$d; mkdir -p $out
cd $GRAFT_REPO_ROOT
run() {  # label, env assignments...
  local label=$1; shift
  local r=$(env "$@" timeout 120 python tools/step_only.py 400 2>&1 | tail -1)
  echo "$label | $r"
}
for rep in 1 2; do
  run base RD_NOP=1
  run dev_kernarg1 HIP_FORCE_DEV_KERNARG=1
  run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0
  run opt_flush0 AMD_OPT_FLUSH=0
  run pkt_capture0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run graph_batch1 DEBUG_HIP_GRAPH_BATCH_SIZE=1
  run graph_batch256 DEBUG_HIP_GRAPH_BATCH_SIZE=256
  run skip_kernarg_copy ROC_SKIP_KERNEL_ARG_COPY=1
  run hwq1 GPU_MAX_HW_QUEUES=1
  run graph_queues1 DEBUG_HIP_FORCE_GRAPH_QUEUES=1
  run sys_scope_sig0 ROC_SYSTEM_SCOPE_SIGNAL=0
  run hsa_no_irq HSA_ENABLE_INTERRUPT=0
  run kernarg_copy_opt0 DEBUG_HIP_KERNARG_COPY_OPT=0
done 2>&1 | tee $out/env_ab.log
