#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/c11; mkdir -p $out; rm -f $out/ab.txt
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_token_plan_gpu.py -m gpu -q -x -k "replayed or matches_padded" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
run() { env "$@" timeout 120 python tools/step_only.py 300 2>&1 | tail -1 | sed "s/^/$* /" >> $out/ab.txt; }
for rep in 1 2 3; do run RD_K1_WARM=1; run RD_K1_WARM=0; done
cat $out/ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in 1 0; do
RD_K1_WARM=$w timeout 120 rocprofv3 --kernel-trace -d $out/kt$w -o step -- python $R/tools/step_only.py 200 > $out/kt$w.log 2>&1
python $R/tools/rocpd_stats.py $(find $out/kt$w -name "*.db" | head -1) 45 | grep "k_dw\|k_msg_bwd\|k_msg_fwd" | sed "s/^/warm=$w /"
rm -rf $out/kt$w
done
