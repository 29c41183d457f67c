"""Host-side helpers of bench.py (no GPU): roofline bookkeeping, PMC traffic file, core count."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_roofline_accounting_matches_design(bench):
    """SURVEY §8d / DESIGN (d): 32*F*K bytes per sample + 24*K^2 (+8K) per launch pair; peak 8 TB/s."""
    B, F, K = 256, 34, 240
    d = bench._roofline_dict(B, F, K, 0.040, 0.070, "unit test")
    assert d["algorithmic_bytes"] == B * 32 * F * K + 24 * K * K + 8 * K == 68231040
    assert d["bound"] == "hbm" and d["unit"] == "GB/s" and d["peak"] == 8000.0
    assert abs(d["achieved"] - 68231040 / 110e-6 / 1e9) < 0.01
    assert abs(d["frac"] - d["achieved"] / d["peak"]) < 1e-4
    assert d["fwd_us"] == 40.0 and d["bwd_us"] == 70.0


def test_pmc_traffic_file_is_consistent(bench):
    path = os.path.join(ROOT, "raindrop_amd", "k1_pmc_traffic.json")
    with open(path) as fh:
        d = json.load(fh)
    assert abs(d["bytes_per_step"] - sum(k["bytes_per_step"] for k in d["kernels"])) <= len(d["kernels"])   # per-kernel rounding
    for k in d["kernels"]:          # FETCH doubled (gfx950 correction), KB -> bytes
        assert abs(k["bytes_per_step"] - k["calls_per_step"] * (2 * k["fetch_kb"] + k["write_kb"]) * 1024) <= 1024
    t, src = bench._pmc_traffic(256, 34, 240)
    if d.get("source_sha1") == bench.k1_source_hash():             # passes taken on these kernel sources
        assert t == d["bytes_per_step"] and "separate passes" in src
        assert t > bench._roofline_dict(256, 34, 240, 0.04, 0.07, "x")["algorithmic_bytes"]
    else:                                                          # kernels edited since: the line must say so, not quote old bytes
        assert t is None and src.startswith("stale")
    assert bench._pmc_traffic(8, 34, 240)[0] is None               # only valid for the shape it was measured on


def test_usable_cores_respects_affinity(bench):
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))


def test_cpu_baseline_is_bounded_by_wall_clock(bench):
    """The CPU leg must never stretch the benchmark: chunks until the sample cap or the wall-clock budget."""
    import time
    from raindrop_amd import synth
    cfg = synth.make_config("P19")
    gs = synth.make_structure(cfg, "ones")
    import torch
    nt = torch.get_num_threads()
    t0 = time.perf_counter()
    try:
        d = bench.cpu_baseline(cfg, gs, 256, 1000, budget_s=2.0)    # cap far beyond the budget
    finally:
        torch.set_num_threads(nt)                                    # the leg picks its own thread count
    wall = time.perf_counter() - t0
    assert d["unit"] == "samples/s" and d["kind"] == "port" and d["value"] > 0
    assert 1 <= d["cores"] <= bench.usable_cores()
    assert "chunks of 32" in d["sample"]
    assert wall < 60.0                                               # probes + warm-up + <= 2 s of timed chunks (+ one chunk)
