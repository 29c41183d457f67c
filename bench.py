#!/usr/bin/env python
"""Benchmark of the Raindrop hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 256] [--config P19]

Step = one training step of `Raindrop_v2` on one synthetic P19-shaped batch of B=256 samples PER
GPU, inputs resident in HBM: forward + CrossEntropyLoss + backward (train mode, dropout 0.2)
[+ flat-gradient all-reduce over RCCL when N>1] + Adam update.  `value` is whole-job samples/s.
For N>1 launch with `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`.

Extra objects on the JSON line (rank 0):
  roofline     -- the fused message-passing kernels (K1, forward+backward) timed with HIP events on
                  the launch stream, against the HBM roofline; algorithmic bytes per SURVEY.md 8(d):
                  B*32*F*K + 24*K^2 per fwd+bwd.
  cpu_baseline -- the CPU port of the reference algorithm (oracle/restatement.py, per-sample /
                  per-edge order) timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ARITH = {
    "bf16x3": "fp32 tensors everywhere; dense contractions as split-bf16 (hi+lo, 3 products) on v_mfma_f32_16x16x32_bf16 with "
              "fp32 accumulation (~2^-16 per product; logits within 1e-4 of the fp32 reference, tests/test_gpu_parity.py); "
              "the attention contractions too (one tile, T <= 64: P19; multi-tile with the scores chained in registers: P12, PAM; "
              "materialised scores for wide heads > 96: SYN256); softmax, LayerNorm, the classifier head and all "
              "reductions in fp32 (RD_PRECISION=fp32 switches every contraction to the exact-fp32 MFMA)",
    "fp32": "fp32 everywhere: dense contractions on v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain)",
    "bf16": "fp32 tensors in HBM; dense contractions with operands rounded to bf16, ONE product per step on "
            "v_mfma_f32_16x16x32_bf16, fp32 accumulation (logits within 3e-2 of the fp32 reference); message passing on the "
            "generic tiled path; softmax, LayerNorm, the classifier head and reductions in fp32",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU")
    ap.add_argument("--config", default="P19")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+CE+bwd(+allreduce) only")
    ap.add_argument("--cpu-reps", type=int, default=64,
                    help="CPU baseline sample cap in batches (it stops after ~15 s of wall clock anyway)")
    ap.add_argument("--k1-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--graph-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--full-graph-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--feed", action="store_true",
                    help="draw every step's batch from a device-resident dataset with rd_batch_gather (SURVEY 8f "
                         "rank 1) instead of re-using one resident batch (the default, as the metric is defined)")
    ap.add_argument("--precision", default=None, choices=["bf16x3", "fp32", "bf16"],
                    help="arithmetic of the dense contractions (default: RD_PRECISION or bf16x3); bf16 = one product, the "
                         "'P12 bf16' configuration of BASELINE.json")
    ap.add_argument("--no-roofline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--lite", action="store_true", help=argparse.SUPPRESS)     # child of other_configs(): the line + its roofline object, no extras
    ap.add_argument("--use-beta", action="store_true",
                    help="the paper's branch (use_beta=True: top-K pruned, per-sample graph) as a captured step: prints tools/bench_use_beta.py's line")
    ap.add_argument("--no-graph", action="store_true",
                    help="eager autograd step instead of the hipGraph-captured static step")
    return ap.parse_args()


def _use_token_plan(cfg):
    """The step runs on the token plan (include/raindrop_hip.h) wherever raindrop_amd.step.TrainStep supports it: the isolated
    roofline loops then measure the kernels on the same layout."""
    from raindrop_amd.step import plan_supported
    prec = {"fp32": 0, "bf16x3": 1, "bf16": 2}[os.environ.get("RD_PRECISION", "bf16x3")]
    D = cfg["d_inp"] * cfg["d_ob"] + 16
    return (os.environ.get("RD_TOKEN_PLAN", "1") != "0"
            and plan_supported(cfg["d_inp"], cfg["d_ob"], cfg["max_len"], D, cfg["nhead"], cfg["nhid"], prec))


def _make_plan(shp, lengths):
    """lengths -> token plan (one launch, outside the timed graphs: it is a function of `lengths` only and shared by every
    kernel of the step)."""
    import ctypes
    from raindrop_amd import _lib
    lib = _lib.load()
    plan = torch.zeros(max(int(lib.rd_token_plan_bytes(ctypes.byref(shp))) // 4, 64), dtype=torch.int32, device=lengths.device)
    _lib.call("rd_token_plan", ctypes.byref(shp), ctypes.c_void_p(lengths.data_ptr()), ctypes.c_void_p(plan.data_ptr()), None, 0,
              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return plan


class _plan_scope:
    """Registers a token plan for the calls enqueued inside (rd_set_token_plan is consumed at enqueue / capture time)."""
    def __init__(self, plan):
        self.plan = plan

    def __enter__(self):
        import ctypes
        from raindrop_amd import _lib
        if self.plan is not None:
            _lib.call("rd_set_token_plan", ctypes.c_void_p(self.plan.data_ptr()))

    def __exit__(self, *a):
        from raindrop_amd import _lib
        _lib.call("rd_set_token_plan", None)


def k1_roofline(model, cfg, batch, reps=20):
    """Time K1 forward and backward in isolation (same tensors the training step uses) with HIP
    events recorded on the stream the kernels are launched on (torch's current stream)."""
    from raindrop_amd import _lib, ops
    dev = batch["src"].device
    B = batch["src"].shape[1]
    T, F, d = cfg["max_len"], cfg["d_inp"], cfg["d_ob"]
    K = T * d
    g = model._graph(dev)
    shp = _lib.shape(B, T, F, d)
    args = (batch["src"], batch["times"], batch["lengths"], model.pos_encoder.timescales(dev), g["ssum"],
            model.R_u, model.ob_propagation.lin_value.weight, model.ob_propagation.lin_value.bias,
            model.ob_propagation_layer2.lin_value.weight, model.ob_propagation_layer2.lin_value.bias, shp,
            0.2, 1234)
    dz = torch.randn(T, B, F * d + 16, device=dev)
    det = [t.detach() for t in args[:10]]
    plan = _make_plan(shp, batch["lengths"]) if _use_token_plan(cfg) else None
    # the raw (autograd-free) entry points the autograd Function itself calls: nothing but the
    # library's launches lands in the captured graph
    def fwd_only():
        with _plan_scope(plan):
            return ops.sensor_stage_fwd_raw(*det, shp, 0.2, 1234)

    def fwd_bwd():
        with _plan_scope(plan):
            z, _, saved = ops.sensor_stage_fwd_raw(*det, shp, 0.2, 1234)
            ops.sensor_stage_bwd_raw(det[0], det[5], det[6], det[8], det[4], saved, z, dz, shp, 0.2)

    INNER = 8

    def time_graph(fn, iters=25):
        """Capture INNER back-to-back calls of `fn` into ONE hipGraph and time replays with HIP events on the replay stream: device
        time of the launches `fn` makes, without host launch gaps -- and with the fixed cost of starting a graph (a few us per replay,
        which the training step pays once for ~30 launches) spread over INNER repetitions instead of charged to 2-5 launches."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(INNER):
                fn()
        for _ in range(3):
            graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (iters * INNER)   # ms per call of fn

    fwd = time_graph(fwd_only)
    both = time_graph(fwd_bwd)
    bwd = both - fwd
    return _roofline_dict(B, F, K, fwd, bwd, "hipGraph replays (8 calls per graph) timed with HIP events" +
                          ("; on the step's token plan (live rows only: %d of %d)" % (int(plan[0]), T * B) if plan is not None else ""))


def k1_in_step(model, flat, batch, iters=40):
    """The K1 launches -- and the encoder layers -- timed INSIDE the training step: the step is captured as six consecutive hipGraphs
    (TrainStep.capture_segments: first launch | message-passing forward | encoder forward | head + loss | encoder backward |
    message-passing backward) and every replay is bracketed by HIP events on the replay stream, while the loop runs whole steps
    back to back.  Unlike the isolated loops (same buffers re-used: resident in the 256 MiB Infinity Cache, which hides exactly the
    HBM traffic the round-4 kernels remove) the kernels see the cache state the step leaves them.  Returns a dict of medians over
    `iters` steps (ms); each interval includes the start of its graph (a few us)."""
    from raindrop_amd.step import TrainStep
    ts = TrainStep(model, flat, batch, use_graph=False, autotune=False)
    parts = ("begin", "k1f", "enc", "head", "encb", "k1b")
    med = lambda v: sorted(v)[len(v) // 2]
    out = None
    try:
        # preferred: ONE graph with external event-record nodes between the parts -- no graph boundary inside the step
        graph, evs = ts.capture_marked(parts)
        for _ in range(5):
            graph.replay()
        torch.cuda.synchronize()
        samples = {p: [] for p in parts}
        steps = []
        for _ in range(iters):
            graph.replay()
            torch.cuda.synchronize()                      # the external events belong to this replay
            for k, p in enumerate(parts):
                samples[p].append(evs[k].elapsed_time(evs[k + 1]))
            steps.append(evs[0].elapsed_time(evs[len(parts)]))
        out = {p: med(samples[p]) for p in parts}
        out["step"] = med(steps)
        out["how"] = "one hipGraph, external event-record nodes between the parts"
        if not all(v > 0 for v in (out[p] for p in parts)):
            out = None
    except Exception as e:                                # the runtime cannot capture external events: segment graphs
        out = None
        sys.stderr.write("k1_in_step: marked capture unavailable (%r): segment graphs\n" % (e,))
    if out is None:
        graphs = ts.capture_segments(parts)
        for _ in range(5):
            for g in graphs:
                g.replay()
        torch.cuda.synchronize()
        n = len(parts)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(n + 1)] for _ in range(iters)]
        for it in range(iters):
            ev[it][0].record()
            for k, g in enumerate(graphs):
                g.replay()
                ev[it][k + 1].record()
        torch.cuda.synchronize()
        out = {p: med([ev[it][k].elapsed_time(ev[it][k + 1]) for it in range(iters)]) for k, p in enumerate(parts)}
        out["step"] = med([ev[it][0].elapsed_time(ev[it][n]) for it in range(iters)])
        # what a graph boundary costs here: the same step as ONE graph (the training configuration), timed the same way; the
        # difference is shared out over the n - 1 extra graph starts and taken off every part but the first
        one = TrainStep(model, flat, batch, use_graph=True, autotune=False)
        for _ in range(5):
            one.run()
        torch.cuda.synchronize()
        e2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for a_, b_ in e2:
            a_.record(); one.run(); b_.record()
        torch.cuda.synchronize()
        g_step = med([a_.elapsed_time(b_) for a_, b_ in e2])
        one.close()
        ovh = max(0.0, (out["step"] - g_step) / n)
        # The parts stay as MEASURED (each interval contains the start of its graph: conservative).  The boundary-corrected values
        # -- a uniform share of (segmented step - one-graph step) taken off every part but the first -- are kept beside them under
        # their own key: round 4 reported those as the headline, and the uniform subtraction over-credits the short segments (a
        # corrected K1 forward came out shorter than the kernel's own minimum duration in the rocprofv3 trace).
        out["corrected"] = {p: (out[p] if p == parts[0] else max(out[p] - ovh, 0.0)) for p in parts}
        out["graph_step"], out["boundary_overhead"] = g_step, ovh
        out["how"] = ("six consecutive hipGraphs, HIP events between their replays, intervals as measured (each contains one graph start; "
                      "external event-record nodes inside one graph are not available on ROCm); (segmented step - the same step as one "
                      "graph) / 6 = %.1f us per boundary here" % (ovh * 1e3))
    out["nl"] = ts.nl
    out["mlive"] = int(ts.plan[0]) if ts.plan is not None else None
    ts.close()
    return out


def encoder_roofline(model, cfg, B, iters=50, lengths=None):
    """Second roofline object: ONE TransformerEncoderLayer forward + backward (rd_encoder_layer_fwd + rd_encoder_layer_bwd
    of layer 0, 13 launches at P19: the other ~85 % of the step) as hipGraph replays timed with HIP events.
    Algorithmic bytes = what the layer must move under its saved-tensor contract, every tensor once per use: forward reads
    x and writes qkv (3D), attention output, the two pre-norm sums, x1, the FFN hidden (nhid) and y; backward reads dy and
    every saved tensor once and writes dx: (18 D + 2 nhid) * 4 bytes per token (weights, statistics and masks are noise)."""
    import ctypes
    from raindrop_amd import _lib, ops
    dev = next(model.parameters()).device
    lib = _lib.load()
    T, F, d = cfg["max_len"], cfg["d_inp"], cfg["d_ob"]
    D, nhid = F * d + 16, cfg["nhid"]
    shp = _lib.shape(B, T, F, d, nhead=cfg["nhead"], nhid=nhid)
    sp = ctypes.byref(shp)
    layer = model.transformer_encoder.layers[0]
    named = dict(layer.named_parameters())
    w = [named[n].detach().contiguous() for n in ops.ENC_PARAM_NAMES]
    g = [torch.empty_like(t) for t in w]
    wp = _lib.RdEncoderPtrs(*[t.data_ptr() for t in w]); gp = _lib.RdEncoderPtrs(*[t.data_ptr() for t in g])
    x = torch.randn(T, B, D, device=dev); y = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    if lengths is None:
        lengths = torch.full((B,), T, dtype=torch.int64, device=dev)
    mask = (torch.arange(T, device=dev)[None, :] >= lengths[:, None]).contiguous()
    saved = torch.zeros(lib.rd_encoder_layer_saved_bytes(sp), dtype=torch.uint8, device=dev)
    ws = torch.zeros(lib.rd_encoder_layer_workspace_bytes(sp), dtype=torch.uint8, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    plan = _make_plan(shp, lengths) if _use_token_plan(cfg) else None

    def fn():
        st = ops._stream()
        with _plan_scope(plan):
            _lib.call("rd_encoder_layer_fwd", sp, 0, P(x), P(mask), ctypes.byref(wp), 0.2, 1234, P(y), P(saved), saved.numel(),
                      P(ws), ws.numel(), st)
            _lib.call("rd_encoder_layer_bwd", sp, 0, P(x), P(mask), ctypes.byref(wp), 0.2, 1234, P(saved), saved.numel(), P(dy),
                      P(dx), ctypes.byref(gp), P(ws), ws.numel(), st)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    for _ in range(5):
        graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    M = T * B
    alg = M * (18 * D + 2 * nhid) * 4
    flops = 3 * 2.0 * M * (3 * D * D + D * D + 2 * D * nhid) + 3 * 4.0 * B * T * T * D     # dense fwd + 2x bwd; attention
    achieved = alg / (ms * 1e-3) / 1e9
    traffic, traffic_src = _enc_pmc_traffic(B, cfg)
    live = int(plan[0]) if plan is not None else M
    return {"bound": "hbm", "kernel": "one TransformerEncoderLayer fwd+bwd (rd_encoder_layer_fwd + rd_encoder_layer_bwd, layer 0 of 2; "
                                       "hipGraph replays timed with HIP events)" +
                                       ("; on the step's token plan: %d live rows of %d" % (live, M) if plan is not None else ""),
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": alg,
            "algorithmic_bytes_live_rows": live * (18 * D + 2 * nhid) * 4,
            "frac_live_rows": round(live * (18 * D + 2 * nhid) * 4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "us": round(ms * 1e3, 2),
            "mfma": {"algorithmic_tflops": round(flops / (ms * 1e-3) / 1e12, 2),
                     "frac_issued": round(_products() * flops / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 5)}}


def roofline_isolated(args):
    """Run the hipGraph-based K1 measurement in a child process so that a capture failure can never
    take the main JSON line down; returns the roofline dict or None."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--k1-child", "--batch", str(args.batch), "--config", args.config]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=240,
                             env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
        out = None
        for ln in res.stdout.splitlines():
            if ln.startswith("K1ROOFLINE "):
                out = json.loads(ln[len("K1ROOFLINE "):])
            if ln.startswith("ENCROOFLINE ") and out is not None:
                out["encoder_layer"] = json.loads(ln[len("ENCROOFLINE "):])
        return out
    except Exception:
        pass
    return None


def k1_roofline_events(model, cfg, batch, reps=20):
    """Fallback: eager launches bracketed by HIP events (includes host launch gaps)."""
    from raindrop_amd import _lib, ops
    dev = batch["src"].device
    B = batch["src"].shape[1]
    T, F, d = cfg["max_len"], cfg["d_inp"], cfg["d_ob"]
    K = T * d
    g = model._graph(dev)
    shp = _lib.shape(B, T, F, d)
    det = [t.detach() for t in (batch["src"], batch["times"], batch["lengths"], model.pos_encoder.timescales(dev),
                                g["ssum"], model.R_u, model.ob_propagation.lin_value.weight,
                                model.ob_propagation.lin_value.bias, model.ob_propagation_layer2.lin_value.weight,
                                model.ob_propagation_layer2.lin_value.bias)]
    dz = torch.randn(T, B, F * d + 16, device=dev)
    ts_f, ts_b = [], []
    for i in range(reps + 3):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        z, _, saved = ops.sensor_stage_fwd_raw(*det, shp, 0.2, 1234)
        e1.record()
        ops.sensor_stage_bwd_raw(det[0], det[5], det[6], det[8], det[4], saved, z, dz, shp, 0.2)
        e2.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts_f.append(e0.elapsed_time(e1)); ts_b.append(e1.elapsed_time(e2))
    fwd, bwd = sorted(ts_f)[len(ts_f) // 2], sorted(ts_b)[len(ts_b) // 2]
    return _roofline_dict(B, F, K, fwd, bwd, "eager launches timed with HIP events (includes host launch gaps)")


def box_kind():
    """Which kind of box ran this line (DESIGN.md "Box variance"): clock64 ticks per 64-byte line of COLD straight-line code (32 KB, one
    wave, right after a 1-GB fill) and of the same code warm.  The pool's fast boxes fetch instructions ahead (~80 cold / ~50 warm),
    the slow ones do not (~400-450 / ~75-95).  Never allowed to cost the line."""
    try:
        import ctypes
        from raindrop_amd import _lib
        lib = _lib.load()
        fn = lib.rd_debug_ifetch_probe
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        fn.restype = None
        t = torch.zeros(1, dtype=torch.int64, device="cuda")
        scr = torch.zeros(64, dtype=torch.float32, device="cuda")
        big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        out = []
        for _ in range(2):
            big.fill_(1.0); torch.cuda.synchronize()
            fn(t.data_ptr(), scr.data_ptr(), st); torch.cuda.synchronize()
            cold = int(t.item())
            fn(t.data_ptr(), scr.data_ptr(), st); torch.cuda.synchronize()
            out.append((cold, int(t.item())))
        del big
        cold, warm = min(o[0] for o in out) / 512.0, min(o[1] for o in out) / 512.0
        return {"cold_code_ticks_per_64B_line": round(cold, 1), "warm_code_ticks_per_64B_line": round(warm, 1),
                "kind": "fast (instruction fetch looks ahead)" if cold < 200 else "slow (cold code is fetched one trip to memory at a time)",
                "how": "32 KB of straight-line dependent v_fma_f32, one wave, clock64; cold = right after a 1-GB fill"}
    except Exception as e:                                               # pragma: no cover
        return {"error": repr(e)[:200]}


K1_SOURCES = ("rd_msgpass_fused.hip", "rd_msgpass_dw.hip", "rd_msgpass.hip", "rd_k1_layout.h")


def k1_source_hash():
    """sha1 over the K1 kernel sources: the committed PMC traffic figure is only valid for the kernels it was measured on."""
    import hashlib
    h = hashlib.sha1()
    for f in K1_SOURCES:
        with open(os.path.join(ROOT, "raindrop_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _k1_rocprof(alg_bytes, shape):
    """The K1 kernels' durations inside the captured step from the committed rocprofv3 kernel trace (tools/k1_rocprof_json.py over
    `rocprofv3 --kernel-trace -- python tools/step_only.py`: the profiler cannot run inside the timed process), valid only for the
    kernel sources it was taken on (sha1 stamp).  This is the figure a reader recomputes from profiles/*_step_kernel_stats.txt."""
    try:
        path = os.path.join(ROOT, "raindrop_amd", "k1_rocprof.json")
        with open(path) as fh:
            d = json.load(fh)
        if list(d.get("shape", [])) != list(shape):
            return None                                   # the trace is of another workload (it is taken on the P19 benchmark batch)
        if d.get("source_sha1") != k1_source_hash():
            return {"stale": "kernel sources changed since the trace was taken", "frac": None}
        us = float(d["k1_us_per_step"])
        return {"us": round(us, 2), "frac": round(alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5), "kernels_us": d.get("kernels_us"),
                "box": d.get("box"), "source": d.get("source")}
    except Exception:
        return None


def trace_kernel_table(db_path):
    """{kernel display name: (calls, average duration in us)} of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace`)."""
    import sqlite3
    c = sqlite3.connect(db_path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % sym)]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    rows = c.execute("select s.%s, count(*), avg(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s"
                     % (name_col, disp, sym, name_col)).fetchall()
    return {name: (calls, avg / 1e3) for name, calls, avg in rows}


# what belongs to K1 in a kernel trace of the captured step: (substring of the kernel name, weight | "share" = K1's share of the step's
# first launch by tile elements, required).  k_dw_reduce is counted WHOLE although it carries encoder layer 0's parked slice reduce
# as a rider (conservative); where the slice sum is folded into the optimizer launch (TrainStep.capture_full, round 6) the launch does
# not exist and K1 is charged the GROWTH of the optimizer launch instead (k1_sum_from_table: adam_base_us).
K1_TRACE_KERNELS = (("k_msg_fwd_fused", 1.0, True), ("k_msg_bwd_fused", 1.0, True), ("k_dw(", 1.0, True), ("k_dw_reduce", 1.0, False),
                    ("k_wsplit", "share", False), ("k_plan", 0.0, False))


def k1_first_launch_share(K=240, D=152, H=272, nl=2):
    k1_el, enc_el = 4 * K * K, nl * 2 * (3 * D * D + D * D + 2 * D * H)
    return k1_el / float(k1_el + enc_el)


def k1_sum_from_table(table, share):
    """us of K1 per step from a kernel table of the captured step + the per-kernel figures that went in."""
    us, total, steps = {}, 0.0, None
    for frag, w, required in K1_TRACE_KERNELS:
        hit = [(n, v) for n, v in table.items() if frag in n]
        if not hit:
            if required:
                raise RuntimeError("kernel %r missing from the trace" % frag)
            continue
        calls, avg = hit[0][1]
        if frag == "k_msg_fwd_fused":
            steps = calls
        us[frag.rstrip("(")] = round(avg, 2)
        total += (share if w == "share" else w) * avg
    return total, us, steps


def live_k1_rocprof(args, cfg, alg_bytes, keep=None):
    """roofline.frac as a MEASUREMENT OF THIS BOX: after the timed region, `rocprofv3 --kernel-trace -- python
    tools/step_only.py` as a child process (the profiler cannot attach to the timed process; the child runs the same captured step on
    the same batch seed, one hipGraph per step incl. Adam), K1's kernel sum read from its rocpd database.  Returns the dict for
    roofline.rocprof or {"error": ..}; never raises.  keep: directory that receives the per-kernel table as text (profiles)."""
    import shutil
    import subprocess
    import tempfile
    if cfg["name"] != "P19":
        return {"error": "tools/step_only.py runs the P19 step only"}
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {"error": "rocprofv3 not on PATH"}
    tmp = tempfile.mkdtemp(prefix="rd_trace_", dir="/tmp")
    try:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        env.update(RD_FULL="1", TMPDIR="/tmp")
        cmd = [exe, "--kernel-trace", "-d", tmp, "-o", "step", "--", sys.executable, os.path.join(ROOT, "tools", "step_only.py"), "100",
               str(args.batch)]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd="/tmp")
        dbs = [os.path.join(r, f) for r, _, fs in os.walk(tmp) for f in fs if f.endswith(".db")]
        if res.returncode != 0 or not dbs:
            return {"error": "rocprofv3 child failed (rc %d): %s" % (res.returncode, (res.stderr or res.stdout)[-300:])}
        table = trace_kernel_table(dbs[0])
        share = k1_first_launch_share(cfg["max_len"] * cfg["d_ob"], cfg["d_inp"] * cfg["d_ob"] + 16, cfg["nhid"], cfg["nlayers"])
        total, us, steps = k1_sum_from_table(table, share)
        step_line = [ln for ln in res.stdout.splitlines() if ln.startswith("loss ")]
        out = {"us": round(total, 2), "frac": round(alg_bytes / (total * 1e-6) / 1e9 / HBM_PEAK_GBS, 5), "kernels_us": us, "steps": steps,
               "k1_share_of_first_launch": round(share, 3),
               "kernel_sum_us_per_step": round(sum(c * a for c, a in table.values() if c >= (steps or 1)) / max(steps or 1, 1), 1),
               "child": step_line[-1] if step_line else None,
               "source": "rocprofv3 --kernel-trace -- python tools/step_only.py 100 %d (RD_FULL=1), spawned by THIS bench.py run after its timed "
                         "region: average kernel durations of the captured step on this box" % args.batch}
        if keep:
            try:
                os.makedirs(keep, exist_ok=True)
                rows = sorted(table.items(), key=lambda kv: -kv[1][0] * kv[1][1])
                tot = sum(c * a for c, a in table.values()) or 1.0
                with open(os.path.join(keep, "bench_step_kernel_stats.txt"), "w") as fh:
                    fh.write("%-92s %7s %12s %10s %6s\n" % ("kernel", "calls", "total_us", "avg_us", "%"))
                    for n, (c, a) in rows[:40]:
                        fh.write("%-92s %7d %12.1f %10.2f %6.2f\n" % (n[:92], c, c * a, a, 100.0 * c * a / tot))
            except Exception:
                pass
        return out
    except Exception as e:
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _pmc_traffic(B, F, K):
    """HBM bytes per K1 step from the committed PMC passes (rocprofv3 --pmc cannot run inside the timed
    process: FETCH_SIZE and WRITE_SIZE need separate passes; tools/k1_traffic_json.py).  Only valid for the
    shape AND the kernel sources it was collected on (sha1 stamp); None otherwise."""
    try:
        path = os.path.join(ROOT, "raindrop_amd", "k1_pmc_traffic.json")
        with open(path) as fh:
            d = json.load(fh)
        if (B, F, K) != (256, 34, 240):
            return None, "PMC passes exist for the P19 B=256 shape only"
        if d.get("source_sha1") != k1_source_hash():
            return None, "stale: the K1 kernel sources changed since the PMC passes (re-run tools/k1_profile.sh pmc + tools/k1_traffic_json.py)"
        return int(d["bytes_per_step"]), d["source"]
    except Exception as e:
        return None, "unavailable: %r" % (e,)


def _enc_pmc_traffic(B, cfg):
    """HBM bytes of one encoder layer fwd+bwd from committed PMC passes over the graph step (raindrop_amd/enc_pmc_traffic.json, made
    by tools/enc_traffic_json.py from separate FETCH_SIZE / WRITE_SIZE runs); valid for the P19 B=256 shape and the kernel sources
    it was collected on."""
    try:
        with open(os.path.join(ROOT, "raindrop_amd", "enc_pmc_traffic.json")) as fh:
            d = json.load(fh)
        if (B, cfg["name"]) != (256, "P19"):
            return None, "PMC passes exist for the P19 B=256 shape only"
        if d.get("source_sha1") != enc_source_hash():
            return None, "stale: the encoder kernel sources changed since the PMC passes (tools/step_pmc.sh + tools/enc_traffic_json.py)"
        return int(d["bytes_per_layer"]), d["source"]
    except Exception as e:
        return None, "unavailable: %r" % (e,)


ENC_SOURCES = ("rd_temporal.hip", "rd_rowgemm.hip", "rd_encfuse.hip", "rd_attnfuse.hip", "rd_tile_wgrad.hip", "rd_plan.hip", "rd_plan.h")


def enc_source_hash():
    import hashlib
    h = hashlib.sha1()
    for f in ENC_SOURCES:
        with open(os.path.join(ROOT, "raindrop_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA


def _products():
    """bf16 MFMA products issued per fp32-equivalent multiply-add in the current arithmetic mode: 3 for split-bf16 (hi*hi + hi*lo +
    lo*hi), 1 for the single-product bf16 mode (the exact-fp32 mode runs on the f32 MFMA: its bf16-peak fraction is reported as if 3)."""
    return 1 if os.environ.get("RD_PRECISION", "bf16x3") == "bf16" else 3


def _roofline_dict(B, F, K, fwd, bwd, how):
    traffic, traffic_src = _pmc_traffic(B, F, K)
    bytes_fwd = B * 12 * F * K + 8 * (K * K + K)
    bytes_bwd = B * 20 * F * K + 16 * K * K
    alg = bytes_fwd + bytes_bwd
    achieved = alg / ((fwd + bwd) * 1e-3) / 1e9
    # 12 F K^2 fp32-equivalent flops per sample fwd+bwd; the split-bf16 path issues 3 MFMA products per flop
    flops = 12.0 * F * K * K * B
    tf = flops / ((fwd + bwd) * 1e-3) / 1e12
    return {"bound": "hbm" if 0.375 * K < 314 else "mfma",
            "kernel": "K1 message passing fwd+bwd (rd_msgpass_fwd + rd_msgpass_bwd incl. PE/mask, "
                      "weight split, dW/db reductions); " + how,
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
            "traffic_note": "FETCH_SIZE counts what the L2s request from the fabric, Infinity-Cache hits included: the 17.8 MB of X / Y1 row tiles "
                            "that k_msg_bwd_fused TOUCHES for the weight-gradient stream behind it (round 4) are counted there AND in k_dw, "
                            "which then finds them in the memory-side cache -- HBM itself still moves them once",
            "algorithmic_bytes": alg, "fwd_us": round(fwd * 1e3, 2), "bwd_us": round(bwd * 1e3, 2),
            "fwd_frac": round(bytes_fwd / (fwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "bwd_frac": round(bytes_bwd / (bwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "mfma": {"algorithmic_tflops": round(tf, 2), "issued_bf16_tflops": round(_products() * tf, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                     "frac_issued": round(_products() * tf / MFMA_BF16_PEAK_TFLOPS, 5), "products_per_flop": _products(),
                     "note": "12 F K^2 flops per sample fwd+bwd (SURVEY 8d); x3 MFMA products in split-bf16 mode; "
                             "arithmetic intensity 0.375 K flop/B: HBM-bound below K ~ 314 with bf16 MFMA, MFMA-bound above"}}


def fp32_mode_ms(args):
    """ms per step of the same training step with the dense contractions on the exact-fp32 MFMA (RD_PRECISION=fp32),
    measured in a child process (the arithmetic mode is process-wide); None if the child fails."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "3", "--batch", str(args.batch), "--config",
           args.config, "--no-cpu-baseline", "--no-roofline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["RD_PRECISION"] = "fp32"
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        for ln in res.stdout.splitlines():
            if ln.startswith("{"):
                return json.loads(ln)["ms_per_step"]
    except Exception:
        pass
    return None


def padded_layout_ms(args):
    """ms per step of the same training step on the PADDED layout (RD_TOKEN_PLAN=0: every sample carried at max_len rows, the
    padded rows computed and masked as in the reference), measured in a child process; None if the child fails."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "20", "--warmup", "5", "--batch", str(args.batch), "--config",
           args.config, "--no-cpu-baseline", "--no-roofline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["RD_TOKEN_PLAN"] = "0"
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        for ln in res.stdout.splitlines():
            if ln.startswith("{"):
                return json.loads(ln)["ms_per_step"]
    except Exception:
        pass
    return None


def recorded_o1(cfg_name, B):
    """The reference's OWN files (oracle O1, under the PyG shim) timed in the build container, train mode, dropout 0.2:
    /root/reference does not exist on the GPU box, so this figure is recorded (raindrop_amd/o1_cpu_baseline.json, made by
    tools/o1_cpu_time.py) rather than measured beside the GPU number."""
    try:
        with open(os.path.join(ROOT, "raindrop_amd", "o1_cpu_baseline.json")) as fh:
            d = json.load(fh)
        return d.get("%s_B%d" % (cfg_name, B))
    except Exception:
        return None


OTHER_CONFIGS = (   # BASELINE.json's other configurations as bounded child runs of this script: (label, argv, seconds)
    ("P12 (36 sensors, T=215) bf16, B=256 [configs[1]]", ["--config", "P12", "--precision", "bf16", "--batch", "256", "--steps", "20", "--warmup", "5"], 200),
    ("PAM (17 sensors, T=600), B=128 [configs[0] shape on the GPU]", ["--config", "PAM", "--batch", "128", "--steps", "10", "--warmup", "3"], 200),
    ("synthetic 256 sensors x 512 steps, B=16 per GPU [configs[4]]", ["--config", "SYN256", "--batch", "16", "--steps", "5", "--warmup", "2"], 240),
)


def other_configs(budget_s=420.0):
    """BASELINE.json names five configurations; the line's metric is quoted on one (P19).  The others run here as CHILD processes
    after the timed region, each bounded, so that the driver's line carries them too (VERDICT r5 #6d): ms/step, samples/s, step mode,
    the message-passing roofline object's MFMA / HBM fractions.  RD_BENCH_OTHER=0 skips them; a child that fails or runs out of time
    leaves an `error` entry, never costs the line."""
    import subprocess
    out, t_start = [], time.perf_counter()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "RD_PRECISION")}
    env.update(RD_BENCH_OTHER="0", RD_BENCH_LIVE_TRACE="0")
    for label, argv, limit in OTHER_CONFIGS:
        left = budget_s - (time.perf_counter() - t_start)
        if left < 30:
            out.append({"config": label, "error": "skipped: the time budget of the other-configuration runs (%.0f s) is spent" % budget_s})
            continue
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--lite"] + argv, capture_output=True,
                                 text=True, timeout=min(limit, left), env=env)
            ln = [x for x in res.stdout.splitlines() if x.startswith("{")]
            if not ln:
                out.append({"config": label, "error": "rc %d: %s" % (res.returncode, (res.stderr or "")[-200:])})
                continue
            d = json.loads(ln[-1])
            rl = d.get("roofline") if isinstance(d.get("roofline"), dict) else {}
            out.append({"config": label, "ms_per_step": d["ms_per_step"], "samples_per_s": d["value"], "dtype": d["dtype"],
                        "step_mode": d["config"].get("step_mode"), "token_plan": str(d["config"].get("token_plan"))[:60],
                        "k1_hbm_frac": rl.get("frac"), "k1_mfma_frac_issued": (rl.get("mfma") or {}).get("frac_issued"),
                        "k1_bound": rl.get("bound")})
        except Exception as e:
            out.append({"config": label, "error": repr(e)[:200]})
    return out


def graph_probe_ok(args, world):
    import subprocess
    if world > 1 and int(os.environ.get("RANK", "0")) != 0:
        pass                                   # every rank probes on its own device: cheap and independent
    cmd = [sys.executable, os.path.abspath(__file__), "--graph-probe", "--batch", str(args.batch), "--config", args.config]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if world > 1 and os.environ.get("RD_BENCH_ONE_GPU") != "1":
        env["HIP_VISIBLE_DEVICES"] = os.environ.get("LOCAL_RANK", "0")
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        return any(ln.startswith("GRAPHPROBE ok") for ln in res.stdout.splitlines())
    except Exception:
        return False


def full_graph_probe_ok(args, world, rank, dev, limit_s=300.0):
    """N > 1, guard (0) of the whole-step graph: RCCL collectives under stream capture between REAL peers have never run on this
    pool (one-rank groups only), and a capture or a replay that HANGS cannot be caught in-process.  So every rank first starts a
    CHILD of this script (`--full-graph-probe`) on its own GPU; the children form their own process group on a fresh port, capture
    the whole step and replay it three times.  A child that fails, or is still running after `limit_s` (killed by exact pid), costs
    the whole-step graph -- all ranks then take the two-graph step with eager collectives -- never the benchmark line."""
    import socket
    import subprocess
    port = torch.zeros(1, dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port[0] = sk.getsockname()[1]
    dist.broadcast(port, 0)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC") and k != "TORCH_NCCL_ASYNC_ERROR_HANDLING"}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(int(port.item())), RD_BENCH_OTHER="0", RD_BENCH_LIVE_TRACE="0")
    cmd = [sys.executable, os.path.abspath(__file__), "--full-graph-probe", "--gpus", str(world), "--batch", str(args.batch),
           "--config", args.config, "--no-cpu-baseline"]
    ok = False
    try:
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        try:
            out, err = child.communicate(timeout=limit_s)
            ok = any(ln.startswith("FULLPROBE ok") for ln in out.splitlines())
            if not ok:
                print("rank %d: whole-step graph probe failed (rc %s): %s" % (rank, child.returncode, (err or "")[-300:]),
                      file=sys.stderr, flush=True)
        except subprocess.TimeoutExpired:
            child.kill()
            child.communicate()
            print("rank %d: whole-step graph probe still running after %.0f s: killed" % (rank, limit_s), file=sys.stderr, flush=True)
    except Exception as e:                                           # pragma: no cover
        print("rank %d: whole-step graph probe could not run (%r)" % (rank, e), file=sys.stderr, flush=True)
    fk = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=port.device)
    dist.all_reduce(fk, op=dist.ReduceOp.MIN)
    return bool(fk.item() > 0.5)


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (a container
    can report 256 CPUs and be throttled to 8; an OpenMP team sized by cpu_count() then crawls)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(cfg, gs, B, reps, budget_s=15.0):
    """CPU port of the reference path in the reference's own evaluation order (per-sample loop,
    per-edge lin_value) -- oracle/restatement.py faithful=True -- forward + CE + backward, on a sample
    bounded by WALL CLOCK: chunks of 32 samples until `reps * B` samples are done or `budget_s` seconds
    have passed, whichever comes first (a probe-sized estimate is not a bound on a noisy host)."""
    from oracle import restatement as O2
    from raindrop_amd import synth
    ncores = usable_cores()
    names = synth.live_parameter_names(cfg)
    import json as _json
    surf = _json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_surface.json")))[cfg["name"]]
    p = {n: synth.param_values(n, surf[n], seed=0).requires_grad_(True) for n in names}
    # thread count: whichever is actually fastest for this evaluation order on a probe (it is thousands of small
    # ops: OpenMP fork/join can make "all cores" slower than a few)
    probe = synth.make_batch(cfg, 8, seed=0)
    best, threads = None, 1
    t_all = time.perf_counter()
    for th in sorted({1, min(4, ncores), min(8, ncores), ncores}):
        if time.perf_counter() - t_all > budget_s / 2:
            break
        torch.set_num_threads(th)
        O2.step_fwd_bwd(p, cfg, probe, gs, faithful=True)
        t0 = time.perf_counter()
        O2.step_fwd_bwd(p, cfg, probe, gs, faithful=True)
        ps = time.perf_counter() - t0
        if best is None or ps < best:
            best, threads = ps, th
    torch.set_num_threads(threads)
    chunk = synth.make_batch(cfg, min(32, max(8, B)), seed=0)
    n_chunk = chunk["src"].shape[1]
    O2.step_fwd_bwd(p, cfg, chunk, gs, faithful=True)                 # warm-up, untimed
    done, t0 = 0, time.perf_counter()
    while done < reps * B and (done == 0 or time.perf_counter() - t0 < budget_s):
        O2.step_fwd_bwd(p, cfg, chunk, gs, faithful=True)
        done += n_chunk
    t = time.perf_counter() - t0
    return {"value": round(done / t, 2), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": "%d %s-shaped samples as fwd+CE+bwd over chunks of %d (after 1 warm-up chunk; bounded by %d samples "
                      "or %.0f s of wall clock), reference evaluation order (per-sample loop, per-edge lin_value), "
                      "dropout off; thread count = fastest of {1,4,8,%d} on a probe"
                      % (done, cfg["name"], n_chunk, reps * B, budget_s, ncores)}


def main():
    import faulthandler
    faulthandler.dump_traceback_later(900, exit=True)      # never hang a GPU box silently
    args = parse()
    if args.use_beta:                                         # a different step (AutogradStep over the composed operators): its own line
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_use_beta
        bench_use_beta.main(["--batch", str(args.batch), "--steps", str(args.steps), "--warmup", str(args.warmup)])
        return
    if args.precision:
        os.environ["RD_PRECISION"] = args.precision           # read by the library at its first call; children inherit it
    prec = os.environ.get("RD_PRECISION", "bf16x3")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d" % (args.gpus, args.gpus))
    # RD_BENCH_ONE_GPU=1 (testing only): every rank shares cuda:0 and talks over gloo, so the
    # multi-rank control flow can be exercised on a 1-GPU box; real runs use one GPU per rank + RCCL.
    one_gpu = os.environ.get("RD_BENCH_ONE_GPU") == "1"
    dev_index = 0 if one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from raindrop_amd import dp, synth
    from raindrop_amd.models_rd import Raindrop_v2

    cfg = synth.make_config(args.config)
    gs = synth.make_structure(cfg, "ones")                       # code/Raindrop.py:212
    torch.manual_seed(1)                                          # code/Raindrop.py:58
    kw = {} if cfg["static"] else {"static": False}
    model = Raindrop_v2(cfg["d_inp"], cfg["d_model"], cfg["nhead"], cfg["nhid"], cfg["nlayers"], cfg["dropout"],
                        cfg["max_len"], cfg["d_static"], cfg["MAX"], 0.5, cfg["aggreg"], cfg["n_classes"], gs,
                        sensor_wise_mask=False, **kw).to(dev)
    if world > 1:
        dp.broadcast_parameters(model, src=0)
    model.train()
    model.graph_step = False    # the timed step below is raindrop_amd.step.TrainStep; `eager_step` = the operator-by-operator module surface (the
                                # module's DEFAULT, the captured step behind model.forward, is measured separately at the end: config.eager_ms_per_step)
    B = args.batch
    batch = synth.make_batch(cfg, B, seed=100 + rank)             # each rank its own shard (weak scaling)
    batch = {k: (None if v is None else v.to(dev)) for k, v in batch.items()}
    live = synth.live_parameter_names(cfg)
    named = dict(model.named_parameters())
    flat = dp.FlatGradAllReduce([(n, named[n]) for n in live], n_buckets=2)
    # code/Raindrop.py:256 (Adam, lr 1e-4) over the live parameters, held in one flat buffer
    from raindrop_amd.optim import FlatAdam
    opt = FlatAdam(flat.flatten_parameters(), lr=1e-4)
    criterion = torch.nn.CrossEntropyLoss()

    # --feed: every step first gathers a fresh batch INTO the resident buffers (one extra launch) from a
    # device-resident dataset, the way code/Raindrop.py:310-317 slices its training tensors on the host
    feed_next = None
    if args.feed and not args.k1_child and not args.graph_probe and not args.full_graph_probe:
        from raindrop_amd import feed as rfeed
        n_feed = 8192
        big = synth.make_batch(cfg, n_feed, seed=200 + rank)
        dset = rfeed.DeviceDataset(big["src"], big["times"], big["static"], big["y"], device=dev)
        gen = torch.Generator().manual_seed(rank)
        idxs = [torch.randint(0, n_feed, (B,), generator=gen).to(dev) for _ in range(64)]
        fbuf = {"P": batch["src"], "Ptime": batch["times"], "Pstatic": batch["static"], "y": batch["y"],
                "lengths": batch["lengths"]}
        fcount = [0]

        def feed_next():
            dset.batch(idxs[fcount[0] % len(idxs)], out=fbuf)
            fcount[0] += 1

    def eager_step():
        if feed_next is not None:
            feed_next()
        flat.zero()
        out, _, _ = model(batch["src"], batch["static"], batch["times"], batch["lengths"])
        loss = criterion(out, batch["y"])
        loss.backward()
        flat.finish()
        if not args.no_optimizer:
            opt.step()
        return loss

    # Static step: fwd + CE + bwd captured once as a hipGraph (raindrop_amd/step.py), gradients written
    # straight into the flat buffer; the all-reduce and the Adam kernel follow eagerly.  A child process
    # first proves that capture + replay work on this box, so a capture failure can only cost the graph,
    # never the benchmark line.
    tstep = None
    use_graph = not args.no_graph and not args.k1_child
    if args.graph_probe:
        from raindrop_amd.step import TrainStep
        ts_ = TrainStep(model, flat, batch)
        for _ in range(3):
            ts_.run()
        torch.cuda.synchronize()
        print("GRAPHPROBE ok %.6f" % float(ts_.loss), flush=True)
        return
    if args.full_graph_probe:                  # child of full_graph_probe_ok: own process group (the parent's ranks, a fresh port)
        from raindrop_amd.step import TrainStep
        ts_ = TrainStep(model, flat, batch)
        ts_.capture_full(opt)
        for _ in range(3):
            l_ = ts_.run_full()
        torch.cuda.synchronize()
        fin = bool(torch.isfinite(l_).item())
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        print("FULLPROBE %s %.6f" % ("ok" if fin else "nonfinite", float(l_)), flush=True)
        return
    probe_ok = bool(use_graph and graph_probe_ok(args, world))
    if world > 1:                               # the step mode is a collective decision: every rank must take
        pk = torch.tensor([1.0 if probe_ok else 0.0], dtype=torch.float64, device=dev)   # the same branches below
        dist.all_reduce(pk, op=dist.ReduceOp.MIN)
        probe_ok = bool(pk.item() > 0.5)
    if probe_ok:
        from raindrop_amd.step import TrainStep
        tstep = TrainStep(model, flat, batch)

    # The whole step -- forward + loss + backward, the all-reduce(s) and Adam -- as ONE hipGraph (TrainStep.capture_full): one replay
    # per step on the host.  Round 5: default on one GPU, opt-in at N > 1.  Round 6: the default at every N (RD_STEP_FULL=0: the
    # two-graph step with eager collectives + the Adam launch) -- behind three guards, because RCCL under stream capture has only run
    # on a one-rank group here (tests/test_dp_gpu.py): (0) children of the ranks try capture + replay first in a group of their own
    # (full_graph_probe_ok: a HANG there is killed after 300 s); (1) the capture is tried per rank and the outcome agreed by an all-reduce(MIN),
    # so either every rank replays the whole-step graph or none does; (2) the first replays are checked: three whole-step replays
    # must leave finite losses on every rank, else all ranks drop back to the two-graph form (the parameters are restored first).
    full_graph = [False]
    # (only RCCL collectives can be captured: the gloo group of the RD_BENCH_ONE_GPU test mode synchronises on the host)
    can_capture = world == 1 or dist.get_backend() == "nccl"
    want_full = tstep is not None and not args.no_optimizer and os.environ.get("RD_STEP_FULL", "1") == "1"
    if world > 1 and want_full and (can_capture or os.environ.get("RD_BENCH_FORCE_FULL_PROBE") == "1"):
        can_capture = full_graph_probe_ok(args, world, rank, dev) and can_capture      # guard (0): hangs are caught in a child
    if tstep is not None and not args.no_optimizer and can_capture and os.environ.get("RD_STEP_FULL", "1") == "1":
        snap = None
        try:
            if world > 1:
                snap = (opt.param.detach().clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.t, tstep.seed_cell.clone())
            tstep.capture_full(opt)
            full_graph[0] = True
        except Exception as e:                                   # pragma: no cover
            print("capture_full failed (%r): two-part step" % (e,), file=sys.stderr, flush=True)
        if world > 1:                                            # a collective decision, like the probe above
            fk = torch.tensor([1.0 if full_graph[0] else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(fk, op=dist.ReduceOp.MIN)
            full_graph[0] = bool(fk.item() > 0.5)
            if full_graph[0]:                                    # guard (2): probe replays
                ok = 1.0
                try:
                    for _ in range(3):
                        l_ = tstep.run_full()
                    torch.cuda.synchronize()
                    ok = 1.0 if bool(torch.isfinite(l_).item()) else 0.0
                except Exception as e:                           # pragma: no cover
                    print("whole-step graph replay failed (%r): two-part step" % (e,), file=sys.stderr, flush=True)
                    ok = 0.0
                fk = torch.tensor([ok], dtype=torch.float64, device=dev)
                dist.all_reduce(fk, op=dist.ReduceOp.MIN)
                if fk.item() < 0.5:
                    full_graph[0] = False
                    with torch.no_grad():
                        opt.param.data.copy_(snap[0]); opt.exp_avg.copy_(snap[1]); opt.exp_avg_sq.copy_(snap[2])
                    opt.t = snap[3]; tstep.seed_cell.copy_(snap[4]); opt._cell_stale = True

    def graph_step():
        if feed_next is not None:
            feed_next()
        if full_graph[0]:
            return tstep.run_full()
        loss = tstep.run_allreduce()           # N > 1: the last layer's + head's gradients are all-reduced beside the rest of the backward
        if not args.no_optimizer:
            opt.step()
        return loss

    def time_mode(fn, n=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    step = eager_step
    t_eager = t_graph = None
    if tstep is not None:
        # keep the graph only if it is actually faster here (e.g. two processes sharing one GPU in the
        # RD_BENCH_ONE_GPU test replay graphs pathologically slowly; one GPU per rank does not)
        t_eager, t_graph = time_mode(eager_step), time_mode(graph_step)
        if world > 1:
            tt = torch.tensor([t_eager, t_graph], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_eager, t_graph = float(tt[0]), float(tt[1])
        if os.environ.get("RD_BENCH_DEBUG_TIMES"):
            print("rank %d: eager %.3f ms, graph %.3f ms" % (rank, t_eager * 1e3, t_graph * 1e3), file=sys.stderr, flush=True)
        if t_graph <= t_eager and os.environ.get("RD_BENCH_FORCE_EAGER") != "1":
            step = graph_step
        else:
            tstep.close(); tstep = None
    else:
        t_eager = time_mode(eager_step)

    if args.k1_child:      # isolated process: K1 roofline only (hipGraph replays), one JSON object on stdout
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        k1 = k1_roofline(model, cfg, batch)
        try:
            # the line's `frac` is the IN-STEP figure: the isolated loop's becomes `frac_isolated`
            seg = k1_in_step(model, flat, batch)
            f_ms, b_ms, begin_ms, seg_step_ms = seg["k1f"], seg["k1b"], seg["begin"], seg["step"]
            # the step's first launch splits the encoder's weights as well: K1's share of it by tile elements
            Kk = cfg["max_len"] * cfg["d_ob"]
            Dd, Hh = cfg["d_inp"] * cfg["d_ob"] + 16, cfg["nhid"]
            k1_el = 4 * Kk * Kk
            enc_el = cfg["nlayers"] * 2 * (3 * Dd * Dd + Dd * Dd + 2 * Dd * Hh)
            share = k1_el / float(k1_el + enc_el)
            us = (f_ms + b_ms + share * begin_ms) * 1e3
            corr = seg.get("corrected")
            us_corr = (corr["k1f"] + corr["k1b"] + share * corr["begin"]) * 1e3 if corr else None
            k1["isolated"] = {"frac": k1["frac"], "achieved": k1["achieved"], "fwd_us": k1["fwd_us"], "bwd_us": k1["bwd_us"],
                              "how": "hipGraph replays of the K1 calls alone (8 per graph), same buffers every call: a ~108 MB working "
                                     "set that stays in the 256 MiB Infinity Cache"}
            k1["frac_isolated"] = k1["frac"]
            # Three in-step figures, all on the line.  `frac` is the one a reader can recompute from the committed kernel trace
            # (profiles/r05c_step_kernel_stats_final.txt): the rocprofv3 kernel sum -- the profiler cannot run inside this process, so it
            # comes from raindrop_amd/k1_rocprof.json, valid only while the kernel sources are the ones it was taken on (sha1 stamp).
            # The LIVE figures of this run are the HIP-event ones: `frac_events` = the K1 segments as measured (each interval
            # contains the start of a hipGraph, ~10 us: conservative), `frac_events_boundary_corrected` = with (segmented step -
            # one-graph step) / 6 taken off every part but the first (round 4's headline; over-credits short segments).  With a stale
            # trace `frac` falls back to the boundary-corrected event figure and says so in `frac_source`.
            fr_ev = round(k1["algorithmic_bytes"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
            fr_corr = round(k1["algorithmic_bytes"] / (us_corr * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if us_corr else None
            rp = _k1_rocprof(k1["algorithmic_bytes"], (args.batch, cfg["d_inp"], cfg["max_len"] * cfg["d_ob"]))
            k1["frac_events"], k1["frac_events_boundary_corrected"] = fr_ev, fr_corr
            if rp:
                k1["rocprof"] = rp
                k1["frac_rocprof"] = rp.get("frac")
            if rp and rp.get("frac"):
                k1["frac"], k1["frac_source"] = rp["frac"], "rocprofv3 kernel sum of the captured step (raindrop_amd/k1_rocprof.json, sources unchanged since)"
                k1["achieved"] = round(k1["algorithmic_bytes"] / (rp["us"] * 1e-6) / 1e9, 2)
            else:
                k1["frac"] = fr_corr if fr_corr else fr_ev
                k1["frac_source"] = "HIP events of this run" + (", graph boundaries calibrated out" if fr_corr else "") + " (no fresh rocprofv3 trace of these sources)"
                k1["achieved"] = round(k1["frac"] * HBM_PEAK_GBS, 2)
            k1["fwd_us"], k1["bwd_us"] = round(f_ms * 1e3, 2), round(b_ms * 1e3, 2)
            k1["in_step"] = {"fwd_us": round(f_ms * 1e3, 2), "bwd_us": round(b_ms * 1e3, 2), "first_launch_us": round(begin_ms * 1e3, 2),
                             "k1_share_of_first_launch": round(share, 3), "us": round(us, 2),
                             "segmented_step_us": round(seg_step_ms * 1e3, 2), "how": seg["how"],
                             "one_graph_step_us": round(seg.get("graph_step", 0.0) * 1e3, 2),
                             "graph_boundary_us": round(seg.get("boundary_overhead", 0.0) * 1e3, 2),
                             "boundary_corrected_segments_us": {p: round(v * 1e3, 2) for p, v in (seg.get("corrected") or {}).items()},
                             "segments_us": {p: round(seg[p] * 1e3, 2) for p in ("begin", "k1f", "enc", "head", "encb", "k1b")}}
            k1["encoder_in_step"] = {"us_per_layer": round((seg["enc"] + seg["encb"]) * 1e3 / seg["nl"], 2), "mlive": seg["mlive"], "nl": seg["nl"],
                                     "us_per_layer_boundary_corrected": (round((corr["enc"] + corr["encb"]) * 1e3 / seg["nl"], 2) if corr else None)}
            k1["kernel"] = ("K1 message passing fwd+bwd AS THEY RUN IN THE TRAINING STEP (rd_sensor_stage_fwd + rd_msgpass_bwd incl. PE/mask, dW/db "
                            "reductions, + K1's share of the step's first launch = its weight split): HIP events between the parts of the "
                            "captured step (" + seg["how"] + "), medians over 40 whole steps; `isolated` is the round-1..3 figure")
        except Exception as e:                                   # the isolated figure survives
            k1["in_step_error"] = repr(e)[:300]
        print("K1ROOFLINE " + json.dumps(k1), flush=True)
        try:
            enc = encoder_roofline(model, cfg, args.batch, lengths=batch["lengths"])
            eis = k1.get("encoder_in_step")
            if eis:
                # the object's time and fractions are the IN-STEP ones; the isolated loop's (Infinity-Cache-resident) stay alongside
                enc["isolated"] = {"us": enc["us"], "frac": enc["frac"], "frac_live_rows": enc["frac_live_rows"]}
                us = eis["us_per_layer"]
                enc["us"] = us
                enc["achieved"] = round(enc["algorithmic_bytes"] / (us * 1e-6) / 1e9, 2)
                enc["frac"] = round(enc["achieved"] / HBM_PEAK_GBS, 5)
                enc["frac_live_rows"] = round(enc["algorithmic_bytes_live_rows"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
                enc["kernel"] = ("one TransformerEncoderLayer fwd+bwd AS IT RUNS IN THE TRAINING STEP: (encoder-forward part + encoder-backward "
                                 "part of the step) / %d layers, HIP events (%s), medians over 40 steps; on the step's token plan: %s live "
                                 "rows; `isolated` = layer 0 alone as hipGraph replays on re-used buffers"
                                 % (eis["nl"], k1["in_step"]["how"], eis["mlive"]))
            if enc["algorithmic_bytes_live_rows"] != enc["algorithmic_bytes"]:
                # token plan on: the kernels touch the live rows only -- `frac` is against THOSE bytes; the padded-layout
                # figure stays as frac_padded_layout
                enc["frac_padded_layout"] = enc["frac"]
                enc["frac"] = enc["frac_live_rows"]
                enc["achieved"] = round(enc["algorithmic_bytes_live_rows"] / (enc["us"] * 1e-6) / 1e9, 2)
            print("ENCROOFLINE " + json.dumps(enc), flush=True)
        except Exception as e:                                   # the K1 object must survive a failure here
            print("ENCROOFLINE_FAILED %r" % (e,), file=sys.stderr, flush=True)
        return

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(loss).item()
    # host side of a step: what this process spends ENQUEUEING one step (graph replay(s), all-reduce calls and waits, the Adam launch),
    # measured as 40 steps issued back to back without a synchronisation in between (the device runs behind; the queue absorbs them)
    torch.cuda.synchronize()
    th0 = time.perf_counter()
    for _ in range(40):
        step()
    host_us = (time.perf_counter() - th0) / 40 * 1e6
    torch.cuda.synchronize()
    feed_check = None
    if feed_next is not None and tstep is not None and world == 1:
        # --feed: the captured graph has just been replayed on a new batch every step.  Check that path against an eager recomputation:
        # a second captured step with dropout OFF is replayed on 4 further batches and every loss compared with the eager model's
        # (eval mode: padded layout, autograd surface) on the same buffers.
        from raindrop_amd.step import TrainStep
        chk = TrainStep(model, flat, batch, p_drop=0.0, autotune=False)
        worst = 0.0
        model.eval()
        for _ in range(4):
            feed_next()
            lg = float(chk.run())
            with torch.no_grad():
                out, _, _ = model(batch["src"], batch["static"], batch["times"], batch["lengths"])
                le = float(criterion(out, batch["y"]))
            worst = max(worst, abs(lg - le) / max(1.0, abs(le)))
        model.train()
        chk.close()
        assert worst < 2e-5, "--feed: replayed graph loss differs from the eager recomputation by %.3g" % worst
        feed_check = {"batches": 4, "max_rel_loss_diff_vs_eager": worst}

    # the unchanged loop with the captured module step (raindrop_amd/graph_module.py, RD_MODULE_GRAPH=1): model.forward -> criterion
    # -> loss.backward() -> flat.finish() + Adam, the model's forward and backward as two hipGraphs behind the nn.Module surface.
    # Measured after the timed region, same batch; never allowed to cost the line.
    t_module = None
    if world == 1 and not args.no_roofline and not args.lite:
        try:
            model.graph_step = None                                       # the module's default: what an unchanged loop gets
            os.environ.pop("RD_MODULE_GRAPH", None)
            eager_step(); eager_step()
            if getattr(model, "_graph_runners", None) and any(r for r in model._graph_runners.values()):
                t_module = time_mode(eager_step, n=20)
        except Exception as e:                                           # pragma: no cover
            print("MODULE_GRAPH_FAILED %r" % (e,), file=sys.stderr, flush=True)
        finally:
            model.graph_step = False
            model.__dict__.pop("_graph_runners", None)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        metric = ("samples/sec fwd+bwd, P19 34-sensor batch=256; % HBM roofline on msg-pass kernel" if (cfg["name"], B) == ("P19", 256)
                  else "samples/sec fwd+bwd, %s %d-sensor batch=%d; %% HBM roofline on msg-pass kernel" % (cfg["name"], cfg["d_inp"], B))
        token_plan_on = tstep is not None and getattr(tstep, "plan", None) is not None
        line = {
            "metric": metric,
            "value": round(world * B * args.steps / elapsed, 1), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if prec == "bf16" else "f32",
            "data": "synthetic",
            "config": {"workload": "%s-shaped synthetic batch (F=%d sensors, T=%d steps, d_ob=4, K=%d), "
                                   "B=%d samples per GPU, Setting-1 (global_structure=ones); step = fwd+CE+bwd"
                                   "%s%s, train mode dropout %.1f" % (
                                       cfg["name"], cfg["d_inp"], cfg["max_len"], cfg["max_len"] * 4, B,
                                       "+RCCL flat-grad all-reduce" if world > 1 else "",
                                       "" if args.no_optimizer else "+Adam", cfg["dropout"]),
                       "step_mode": ("eager autograd" if tstep is None else
                                     "ONE hipGraph per step: fwd+CE+bwd%s + Adam (device-side step state)" % (
                                         ", both all-reduce buckets captured (the first beside the rest of the backward)" if world > 1 else "")
                                     if full_graph[0] else
                                     "two hipGraphs (fwd+CE+bwd of head and last layer | rest of bwd), first all-reduce bucket between them, + Adam"
                                     if tstep.split else "hipGraph(fwd+CE+bwd) + eager all-reduce/Adam"),
                       "batch_source": ("rd_batch_gather from a device-resident dataset (N=8192) every step" if feed_next
                                        else "one resident batch re-used (inputs in HBM before the timed region)"),
                       "arithmetic": ARITH[prec],
                       # the drop-in path of code/Raindrop.py:319-323 (model.forward -> criterion -> loss.backward() through autograd, +
                       # flat.finish() + Adam), same batch, same kernels: what the unmodified script gets BY DEFAULT -- since round 5 the
                       # model's forward / backward as two hipGraphs behind the nn.Module surface (raindrop_amd/graph_module.py; the
                       # loss, autograd's accumulation into p.grad and the optimizer stay the loop's)
                       # (`eager_ms_per_step` keeps its rounds 1-4 meaning -- the loop operator by operator under autograd,
                       # RD_MODULE_GRAPH=0 -- and the module's default is reported under its own keys)
                       "eager_ms_per_step": None if t_eager is None else round(t_eager * 1e3, 4),
                       "operator_by_operator_ms_per_step": None if t_eager is None else round(t_eager * 1e3, 4),
                       "module_default_ms_per_step": None if t_module is None else round(t_module * 1e3, 4),
                       "module_graph_ms_per_step": None if t_module is None else round(t_module * 1e3, 4),
                       "token_plan": ("on: the padding mask (code/models_rd.py:298-299) applied as a layout -- only the %d live (sample, step) rows "
                                      "of %d are stored and processed; logits, loss and every gradient are the same function of the inputs "
                                      "(tests/test_token_plan_gpu.py); config.padded_layout_ms_per_step is the same step with every padded row "
                                      "computed and masked" % (int(tstep.plan[0]), B * cfg["max_len"])) if token_plan_on else "off (padded layout)",
                       "feed_check": feed_check,
                       "tuned": (None if tstep is None else {"rowgemm_rows32": tstep.tuned_rows32, "rowgemm_waves16": tstep.tuned_waves16,
                                                             "how": "capture-time A/B of the row-block kernels' workgroup shapes, per-variant "
                                                                    "times summed over the ranks (every rank runs the same variants)"}),
                       "host_us_per_step": round(host_us, 1),
                       "global_batch": world * B, "parallelism": "dp%d" % world,
                       "grad_allreduce_bytes": flat.nbytes()},
        }
        # K1 roofline: rank 0's own launches (hipGraph replays in an isolated child process, else HIP events in
        # this one), after the timed region -- the other ranks are idle by then; never allowed to cost the line
        if not args.no_roofline:
            try:
                line["roofline"] = roofline_isolated(args) or k1_roofline_events(model, cfg, batch)
                if isinstance(line["roofline"], dict) and "encoder_layer" in line["roofline"]:
                    line["roofline_encoder_layer"] = line["roofline"].pop("encoder_layer")   # second object: the other ~85 % of the step
                rl = line["roofline"]
                if world == 1 and isinstance(rl, dict) and "algorithmic_bytes" in rl and os.environ.get("RD_BENCH_LIVE_TRACE", "1") != "0":
                    # frac = the rocprofv3 kernel sum of the captured step ON THIS BOX, taken by a child of this run; the committed
                    # trace (raindrop_amd/k1_rocprof.json, another box) stays as frac_stamped; if the child fails frac is the
                    # CONSERVATIVE live figure (HIP events, graph starts included), not the boundary-corrected one
                    live = live_k1_rocprof(args, cfg, rl["algorithmic_bytes"], keep=os.environ.get("RD_BENCH_KEEP_TRACE"))
                    rl["frac_stamped"] = rl.pop("frac_rocprof", None)
                    rl["rocprof_stamped"] = rl.pop("rocprof", None)
                    rl["rocprof"] = live
                    if live.get("frac"):
                        rl["frac"], rl["achieved"] = live["frac"], round(live["frac"] * HBM_PEAK_GBS, 2)
                        rl["frac_source"] = "rocprofv3 --kernel-trace child of THIS run (roofline.rocprof): K1 kernel sum of the captured step on this box"
                    elif rl.get("frac_events"):
                        rl["frac"], rl["achieved"] = rl["frac_events"], round(rl["frac_events"] * HBM_PEAK_GBS, 2)
                        rl["frac_source"] = "HIP events of this run, graph starts included (conservative): the rocprofv3 child failed: %s" % live.get("error")
            except Exception as e:                                       # pragma: no cover
                line["roofline"] = {"error": repr(e)[:200]}
            if world == 1:
                line["config"]["box"] = box_kind()
                if isinstance(line.get("roofline"), dict) and isinstance(line["roofline"].get("rocprof"), dict):
                    line["roofline"]["rocprof"]["box"] = line["config"]["box"]      # the trace is of THIS box
                if not args.lite:
                    line["config"]["fp32_mode_ms_per_step"] = fp32_mode_ms(args)
                    if token_plan_on:
                        line["config"]["padded_layout_ms_per_step"] = padded_layout_ms(args)
                    if (cfg["name"], B) == ("P19", 256) and os.environ.get("RD_BENCH_OTHER", "1") != "0":
                        line["config"]["other_configs"] = other_configs()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, gs, B, args.cpu_reps)
            line["cpu_baseline"]["reference_o1"] = recorded_o1(cfg["name"], B)
            # what the baseline IS, where a reader of `config` sees it: the restatement (port) with dropout OFF on this
            # host's cores; the reference's own files (O1, dropout 0.2) cannot run here (/root/reference is not on the GPU box) and
            # their figure is a recording from the build container
            line["config"]["cpu_baseline_kind"] = ("port (oracle/restatement.py, reference evaluation order), dropout off, %d threads of THIS host; "
                                                   "cpu_baseline.reference_o1 = the reference's own files, dropout 0.2, RECORDED in the build "
                                                   "container (8 cores), not measured here" % line["cpu_baseline"]["cores"])
            line["cpu_baseline"]["kind"] = "port"
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                           # rank 0 measured the roofline after the timed region: tear down together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
