#!/usr/bin/env python
"""The K1 kernels' durations INSIDE the captured training step from a rocprofv3 kernel trace of tools/step_only.py, as the JSON that
bench.py reports as roofline.rocprof / roofline.frac_rocprof (the figure a reader recomputes from profiles/*_step_kernel_stats.txt):

    rocprofv3 --kernel-trace -d DIR -o step -- python tools/step_only.py 200
    python tools/k1_rocprof_json.py DIR/.../step_results.db [box-kind json] > raindrop_amd/k1_rocprof.json

Per step: k_msg_fwd_fused + k_msg_bwd_fused + k_dw + k_dw_reduce (which carries encoder layer 0's parked slice reduce as a rider:
counted whole, conservative) + K1's share (by tile elements) of the step's first launch (k_wsplit: the weight splits of the whole
step + the token plan)."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(path, box=None):
    import bench
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    rows = c.execute("select s.%s, count(*), avg(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s"
                     % (name_col, disp, sym, name_col)).fetchall()
    pick = {"k_msg_fwd_fused": None, "k_msg_bwd_fused": None, "k_dw(": None, "k_dw_reduce": None, "k_wsplit": None}
    for name, calls, avg in rows:
        for k in pick:
            if k in name:
                pick[k] = (calls, avg / 1e3)
    missing = [k for k, v in pick.items() if v is None]
    if missing:
        raise SystemExit("kernels missing from the trace: %s" % missing)
    # K1's share of the first launch by tile elements, as bench.py computes it for the P19 shape
    K, D, H, nl = 240, 152, 272, 2
    k1_el, enc_el = 4 * K * K, nl * 2 * (3 * D * D + D * D + 2 * D * H)
    share = k1_el / float(k1_el + enc_el)
    us = {k.rstrip("("): round(v[1], 2) for k, v in pick.items()}
    total = pick["k_msg_fwd_fused"][1] + pick["k_msg_bwd_fused"][1] + pick["k_dw("][1] + pick["k_dw_reduce"][1] + share * pick["k_wsplit"][1]
    out = {"source_sha1": bench.k1_source_hash(), "shape": [256, 34, K],          # (B, F, K) of tools/step_only.py: the P19 benchmark batch
           "source": "rocprofv3 --kernel-trace over tools/step_only.py (the captured training step, P19 B=256): average kernel durations, "
                     "%d steps; k_dw_reduce counted whole (it carries encoder layer 0's slice reduce as a rider); k_wsplit x %.3f" % (
                         pick["k_msg_fwd_fused"][0], share),
           "k1_us_per_step": round(total, 2), "kernels_us": us, "k1_share_of_first_launch": round(share, 3)}
    if box:
        try:
            out["box"] = json.loads(open(box).read().split("BOX", 1)[1])
        except Exception:
            pass
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
