"""Which kind of box is this (DESIGN.md "Box variance")?  Prints bench.py's box_kind() record on one line.  First line of every GPU call."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
import bench
k = bench.box_kind()
print("BOX", json.dumps({a: k.get(a) for a in ("kind", "cold_code_ticks_per_64B_line", "warm_code_ticks_per_64B_line", "error") if a in k}))
