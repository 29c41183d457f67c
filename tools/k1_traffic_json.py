#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes over tools/k1_only.py (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into the per-step HBM traffic figure bench.py reports as roofline.traffic.

    python tools/k1_traffic_json.py gpurun_out/pmc_f/p_results.db gpurun_out/pmc_w/p_results.db \
        > raindrop_amd/k1_pmc_traffic.json

Per K1 step (one rd_sensor_stage_fwd + one rd_msgpass_bwd): sum over the step's kernels of
calls_per_step * (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes -- the counters report KB, and FETCH_SIZE is
doubled on gfx950 (it counts 64-byte requests as 32) per the guide's correction."""
import json
import re
import sqlite3
import sys
from collections import defaultdict

K1_KERNELS = ("k_wprep", "k_msg_fwd_fused", "k_msg_bwd_fused", "k_dw(", "k_dw_reduce", "k_gemm_bf16x3<false, false", "k_reduce_wide",
              "k_splitk_reduce2", "k_colsum_small", "k_wgrad_slab", "k_pe_mask", "k_obs_embed", "k_msg_dz2")


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    tab = lambda p: [t for t in tabs if t.startswith(p)][0]
    disp, sym, pmc, info = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    q = ("select s.%s, e.value, d.id from %s e join %s i on e.pmc_id = i.id join %s d on e.event_id = d.event_id "
         "join %s s on d.kernel_id = s.id where i.name = ?" % (name_col, pmc, info, disp, sym))
    tot, ids = defaultdict(float), defaultdict(set)
    for name, val, did in c.execute(q, (counter,)):
        tot[name] += val
        ids[name].add(did)
    return {n: (tot[n] / len(ids[n]), len(ids[n])) for n in tot}


def main(fetch_db, write_db, steps=10):
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    rows, total = [], 0.0
    for name in sorted(f):
        if not any(k in name for k in K1_KERNELS):
            continue
        fk, n = f[name]
        wk = w.get(name, (0.0, n))[0]
        calls = n / float(steps)
        b = calls * (2.0 * fk + wk) * 1024.0
        total += b
        m = re.search(r"(k_\w+(?:<[^>]*>)?)", name)
        rows.append({"kernel": m.group(1) if m else name[:80], "calls_per_step": calls,
                     "fetch_kb": round(fk, 1), "write_kb": round(wk, 1), "bytes_per_step": round(b)})
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    print(json.dumps({"source_sha1": bench.k1_source_hash(), "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/k1_only.py; "
                                "FETCH_SIZE doubled for gfx950; P19 shape, B=256",
                      "bytes_per_step": round(total), "kernels": rows}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 10)
