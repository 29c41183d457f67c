#!/bin/bash
# Timing ablations of the fused message-passing kernels: builds variants of the library with phases compiled out (RD_ABL bit mask,
# rd_msgpass_fused.hip) HERE, then on the GPU box:  for m in 0 1 2 ...; do RD_LIB_PATH=raindrop_amd/_ab/lib_abl$m.so python tools/k1_time.py; done
# usage (build container): tools/k1_ablate.sh 1 2 3 4 8 16 31
set -e
cd "$(dirname "$0")/.."
for m in "$@"; do tools/ab_build.sh abl$m raindrop_amd/csrc/rd_msgpass_fused.hip "-DRD_ABL=$m" > /dev/null; echo abl$m; done
