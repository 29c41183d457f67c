#!/bin/bash
# kernel resource usage (VGPR / spills / LDS / occupancy) of one .hip file; also leaves the ISA in /tmp/kres/
# usage: tools/kres.sh raindrop_amd/csrc/rd_msgpass_fused.hip
mkdir -p /tmp/kres && cd /tmp/kres
f=$(readlink -f "$OLDPWD/$1" 2>/dev/null || readlink -f "$1")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I /root/repo/include -c "$f" -o /tmp/kres/out.o --save-temps \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Spill|Occupancy|ScratchSize" \
  | sed -E 's/.*remark: [^ ]+ +//; s/\[-Rpass.*//' | paste - - - - - - | sed -E 's/Function Name: //'
