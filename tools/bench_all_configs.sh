#!/bin/bash
# the bench lines of every configuration (rounds 4-5) (P19 full line; P12 both bf16 modes at B = 256 with rooflines; PAM; SYN256)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd $GRAFT_REPO_ROOT
[ -n "$SKIP_P19" ] || timeout 600 python bench.py --steps 50 --warmup 10 > $out/bench_P19.json 2> $out/bench_P19.err
timeout 600 python bench.py --config P12 --batch 256 --precision bf16 --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_P12_bf16.json 2> $out/bench_P12_bf16.err
timeout 600 python bench.py --config P12 --batch 256 --precision bf16x3 --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_P12_bf16x3.json 2> $out/bench_P12_bf16x3.err
timeout 600 python bench.py --config PAM --batch 64 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_PAM.json 2> $out/bench_PAM.err
timeout 600 python bench.py --config SYN256 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_SYN256.json 2> $out/bench_SYN256.err
for f in P19 P12_bf16 P12_bf16x3 PAM SYN256; do python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$f.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    e=d.get("roofline_encoder_layer") or {}
    print("$f", d["ms_per_step"], d["value"], "K1 frac", r.get("frac"), "iso", r.get("frac_isolated"), "mfma", (r.get("mfma") or {}).get("frac_issued"), "enc us", e.get("us"), "enc frac_live", e.get("frac_live_rows"), "plan:", d["config"]["token_plan"][:12])
except Exception as ex:
    print("$f FAILED", ex)
PY
done
