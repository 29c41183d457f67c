#!/bin/bash
# round 4 final: profile set (K1-only + step kernel traces, FETCH / WRITE / SQ PMC passes, traffic JSONs) and every config's bench line
d=$1
bash $GRAFT_REPO_ROOT/tools/r4_profiles.sh $d > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp gpurun_out/$d/k1_pmc_traffic.json raindrop_amd/k1_pmc_traffic.json
cp gpurun_out/$d/enc_pmc_traffic.json raindrop_amd/enc_pmc_traffic.json
bash tools/r4_bench_all.sh $d
