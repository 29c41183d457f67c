#!/bin/bash
# Only the encoder's PMC traffic JSON (raindrop_amd/enc_pmc_traffic.json, stamped with the sha1 of the encoder sources): the FETCH_SIZE and
# WRITE_SIZE passes over the captured step, as in tools/r5_profiles.sh -- for a change that touched an encoder source but no kernel
# of the message-passing stage (whose JSONs stay valid).    usage: tools/enc_traffic_refresh.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
R=$GRAFT_REPO_ROOT
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
cd /tmp && export TMPDIR=/tmp
st() { timeout 120 rocprofv3 --kernel-trace --pmc $2 -d $out/st_$1 -o step -- python $R/tools/step_only.py 8 > $out/st_$1.log 2>&1; }
db() { find $out/$1 -name "*.db" | head -1; }
st f FETCH_SIZE; st w WRITE_SIZE
python $R/tools/rocpd_pmc.py $(db st_f) "rd::" > $out/step_pmc_fetch.txt 2>&1
python $R/tools/rocpd_pmc.py $(db st_w) "rd::" > $out/step_pmc_write.txt 2>&1
python $R/tools/enc_traffic_json.py $(db st_f) $(db st_w) > $out/enc_pmc_traffic.json 2> $out/enc_traffic.err
rm -rf $out/st_f $out/st_w
cd $R; cp $out/enc_pmc_traffic.json raindrop_amd/enc_pmc_traffic.json
python -c "import json; d=json.load(open('raindrop_amd/enc_pmc_traffic.json')); print(d['source_sha1'], d['bytes_per_layer'])"
