#!/usr/bin/env python
"""HBM traffic of ONE encoder layer (forward + backward) from the two rocprofv3 PMC passes over the hipGraph training step
(tools/step_pmc.sh: FETCH_SIZE and WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes) -> what bench.py reports as
roofline_encoder_layer.traffic.

    python tools/enc_traffic_json.py gpurun_out/<dir>/f/..._results.db gpurun_out/<dir>/w/..._results.db > raindrop_amd/enc_pmc_traffic.json

Per kernel: average (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per dispatch (the counters report KB; FETCH_SIZE is doubled on gfx950
per the guide's correction) times its dispatches per training step (= its dispatch count / the count of k_msg_fwd_fused, which runs
once per step); the layer figure is the sum over the encoder's kernels divided by the two layers."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from k1_traffic_json import per_kernel                     # noqa: E402

ENC_KERNELS = ("k_enc_post_fwd", "k_enc_pre_bwd", "k_rowgemm", "k_attn_", "k_twg", "k_ln_", "k_add_ln", "k_gemm_bf16x3", "k_splitk", "k_colsum")
SHARED = ("k_wsplit",)                                      # the step's weight splits: the encoder's share by bytes is ~all of it


def main(fetch_db, write_db, nlayers=2):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    steps = [n for name, (_, n) in f.items() if "k_msg_fwd_fused" in name]
    assert steps, "no k_msg_fwd_fused dispatch in the trace: not the training step?"
    nstep = float(sum(steps))
    rows, total = [], 0.0
    for name in sorted(f):
        if not any(k in name for k in ENC_KERNELS + SHARED):
            continue
        fk, n = f[name]
        wk = w.get(name, (0.0, n))[0]
        calls = n / nstep
        b = calls * (2.0 * fk + wk) * 1024.0
        total += b
        m = re.search(r"(k_\w+(?:<[^>]*>)?)", name)
        rows.append({"kernel": m.group(1) if m else name[:80], "calls_per_step": round(calls, 3), "fetch_kb": round(fk, 1),
                     "write_kb": round(wk, 1), "bytes_per_step": round(b)})
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    print(json.dumps({"source_sha1": bench.enc_source_hash(),
                      "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/step_only.py (the hipGraph "
                                "training step on its token plan); FETCH_SIZE doubled for gfx950; encoder kernels of both layers / 2; P19, B=256",
                      "bytes_per_layer": round(total / nlayers), "bytes_per_step_encoder": round(total), "kernels": rows}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
