#!/bin/bash
# Kernel trace of one python target under rocprofv3, summarised by tools/rocpd_stats.py.
#   tools/ktrace.sh <outfile> <rows> [ENV=VALUE ...] -- <python script> [args]
# e.g. tools/ktrace.sh gpurun_out/x/k1_default.txt 8 RD_LIB_PATH=raindrop_amd/_ab/lib_v.so -- tools/k1_only.py 10
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
outf=$1; rows=$2; shift 2
envs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done
shift
case "$outf" in /*) ;; *) outf=$R/$outf;; esac
mkdir -p "$(dirname "$outf")"
tmp=$(mktemp -d /tmp/ktrace.XXXXXX)
args=()
for a in "$@"; do case "$a" in tools/*|bench.py) args+=("$R/$a");; *) args+=("$a");; esac; done
( cd /tmp && export TMPDIR=/tmp && env "${envs[@]/#RD_LIB_PATH=raindrop_amd/RD_LIB_PATH=$R/raindrop_amd}" timeout 240 rocprofv3 --kernel-trace -d $tmp -o t -- python "${args[@]}" > $tmp/log.txt 2>&1 )
db=$(find $tmp -name "*.db" | head -1)
if [ -z "$db" ]; then echo "ktrace: no database; log tail:" > "$outf"; tail -20 $tmp/log.txt >> "$outf"; else python $R/tools/rocpd_stats.py $db $rows > "$outf" 2>&1; fi
rm -rf $tmp
