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
    table = bench.trace_kernel_table(path)
    K = 240
    share = bench.k1_first_launch_share()
    total, us, steps = bench.k1_sum_from_table(table, share)
    out = {"source_sha1": bench.k1_source_hash(), "shape": [256, 34, K],          # (B, F, K) of tools/step_only.py: the P19 benchmark batch
           "source": "rocprofv3 --kernel-trace over tools/step_only.py (the captured training step, P19 B=256): average kernel durations, "
                     "%d steps; k_dw_reduce (where it exists) counted whole (it carries encoder layer 0's slice reduce as a rider); "
                     "first launch x %.3f" % (steps, share),
           "k1_us_per_step": round(total, 2), "kernels_us": us, "k1_share_of_first_launch": round(share, 3)}
    if box:
        try:
            out["box"] = json.loads(open(box).read().split("BOX", 1)[1])
        except Exception:
            pass
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
