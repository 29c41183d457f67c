#!/bin/bash
# Only the two message-passing JSON files bench.py reads (raindrop_amd/k1_rocprof.json: the K1 kernels' durations in the captured step;
# raindrop_amd/k1_pmc_traffic.json: FETCH_SIZE / WRITE_SIZE passes over tools/k1_only.py), stamped with the sha1 of the K1 sources -- for a
# change that touched only those (the encoder's JSON stays valid).  Most important first.   usage: tools/k1_json_refresh.sh <outdir under gpurun_out>
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
R=$GRAFT_REPO_ROOT
python $R/tools/box_kind.py 2>&1 | grep BOX > $out/box.txt; cat $out/box.txt
export RD_RG_ROWS32=15 RD_RG_WAVES16=12
cd /tmp && export TMPDIR=/tmp
db() { find $out/$1 -name "*.db" | head -1; }
timeout 100 rocprofv3 --kernel-trace -d $out/st_kt -o step -- python $R/tools/step_only.py 200 > $out/st_kt.log 2>&1
python $R/tools/rocpd_stats.py $(db st_kt) 45 > $out/step_kernel_stats.txt 2>&1
python $R/tools/k1_rocprof_json.py $(db st_kt) $out/box.txt > $out/k1_rocprof.json 2> $out/k1_rocprof.err
cp $out/k1_rocprof.json $R/raindrop_amd/k1_rocprof.json; rm -rf $out/st_kt
head -13 $out/step_kernel_stats.txt | cut -c1-60,90-150
k1() { timeout 60 rocprofv3 --kernel-trace --pmc $2 -d $out/k1_$1 -o k1 -- python $R/tools/k1_only.py 10 > $out/k1_$1.log 2>&1; }
k1 f FETCH_SIZE; k1 w WRITE_SIZE
python $R/tools/rocpd_pmc.py $(db k1_f) "rd::" > $out/k1_pmc_fetch.txt 2>&1
python $R/tools/rocpd_pmc.py $(db k1_w) "rd::" > $out/k1_pmc_write.txt 2>&1
python $R/tools/k1_traffic_json.py $(db k1_f) $(db k1_w) > $out/k1_pmc_traffic.json 2> $out/k1_traffic.err
rm -rf $out/k1_f $out/k1_w
cp $out/k1_pmc_traffic.json $R/raindrop_amd/k1_pmc_traffic.json
cd $R; python -c "
import json
for f in ('k1_rocprof','k1_pmc_traffic'):
    d=json.load(open('raindrop_amd/'+f+'.json')); print(f, d['source_sha1'][:10], {k:d[k] for k in d if k in ('k1_us_per_step','kernels_us','bytes_per_step')})"
