"""Static training step for `Raindrop_v2`: forward + CrossEntropyLoss + backward as ONE hipGraph.

`code/Raindrop.py:319-323` runs `model.forward -> criterion -> loss.backward()` through autograd,
~100 host-side launches per step.  Every stage of this model already has a forward and a backward
entry point in the C-ABI, so the step can be written out explicitly -- no autograd graph, no
Python in the replay path, every buffer allocated once:

    z, mask          <- rd_sensor_stage_fwd                     (K1 + PE + mask)
    x_{l+1}          <- rd_encoder_layer_fwd(x_l)               (K2/K3, per layer)
    feat = [agg|emb] <- rd_masked_mean_fwd, rd_linear_fwd       (K5; both write into one buffer)
    logits           <- rd_linear_fwd x2
    loss, dlogits    <- rd_softmax_xent                         (CrossEntropyLoss fwd+bwd, one launch)
    ... the same chain backwards, each gradient written straight into its slice of the flat
    gradient buffer (raindrop_amd.dp.FlatGradAllReduce): no accumulate kernels, no packing copy.

The captured graph is replayed per step; dropout masks change per replay through the device seed
cell (`rd_set_seed_cell` / `rd_seed_cell_advance`), which the graph bumps itself.  The gradient
all-reduce (N > 1) and the Adam kernel stay outside the graph.  The eager model (`models_rd.py`)
and this step call the SAME kernels in the same order; `tests/test_gpu_parity.py` checks that the
gradients agree bit for bit.
"""
import ctypes
import os
import time

import torch

from . import _lib, ops


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def plan_supported(d_inp, d_ob, T, D, nhead, nhid, precision):
    """Shapes / modes whose whole step runs on kernels that read a token plan (include/raindrop_hip.h "token plan"): a bf16
    arithmetic mode (precision 1 = bf16x3, 2 = bf16), the fused row-local encoder chains (ceil(D / 32) == 5 and
    ceil(nhid / 32) == 9: the P19 and P12 widths), head_dim <= 96 (single-tile attention for T <= 64 -- with in_proj fused in,
    rd_attnfuse.hip -- or the multi-tile kernels beyond), and either message-passing form (fused LDS-resident for F <= 48,
    K <= 240 in bf16x3, else the panel products whose last scatter follows the plan).  Round 3 had this for the P19 envelope
    only; P12 (T = 215) joined in round 4.  The usual A/B switches of the kernels it relies on turn it off."""
    hd = D // nhead
    env_on = all(os.environ.get(k, "1") != "0" for k in ("RD_ROWGEMM", "RD_TILE_WGRAD", "RD_ATTN_B16", "RD_ATTN_B16_MT", "RD_LN_FUSE",
                                                         "RD_LNB_FUSE", "RD_ENC_FUSE", "RD_HEAD_FUSED"))
    return (precision in (1, 2) and d_ob == 4 and hd * nhead == D and hd <= 96 and (D + 31) // 32 == 5 and (nhid + 31) // 32 == 9
            and D % 4 == 0 and nhid % 4 == 0 and env_on and os.environ.get("RD_ATTN_BIG", "0") == "0")


class TrainStep:
    def __init__(self, model, flat, batch, p_drop=None, use_graph=True, seed=1234, autotune=True, token_plan=None, split=None,
                 module_mode=False):
        """model: raindrop_amd.models_rd.Raindrop_v2 on a ROCm device; flat: FlatGradAllReduce over the
        live parameters (its buffer receives the gradients); batch: dict(src, static, times, lengths, y)
        of device tensors that are REUSED every step (copy new data into them).
        token_plan: store and process only the live (sample, step) rows (include/raindrop_hip.h "token plan": the padding mask of
        code/models_rd.py:298-299 applied as a layout; same logits, loss and gradients).  None = environment RD_TOKEN_PLAN
        (default on) where the shape supports it."""
        self.model, self.flat, self.batch = model, flat, batch
        # module_mode (raindrop_amd.graph_module): the loss is the CALLER's -- the step is cut into parts 'mf' (forward up to the
        # logits) and 'mb' (backward from self.dlogits, which the caller fills), batch carries no labels
        self.module_mode = bool(module_mode)
        # split: capture the step as TWO graphs -- (A) forward + loss + the backward of the head and the last encoder layer, (B) the
        # rest of the backward pass -- so that a data-parallel caller can start the all-reduce of the gradients A has finished
        # (run(between=...)) beside B.  None = on when torch.distributed runs more than one rank (RD_DP_OVERLAP=0 turns it off).
        if split is None:
            import torch.distributed as dist
            split = (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
                     and os.environ.get("RD_DP_OVERLAP", "1") != "0")
        self.split = bool(split) and len(model.transformer_encoder.layers) >= 2
        self._want_plan = (os.environ.get("RD_TOKEN_PLAN", "1") != "0") if token_plan is None else bool(token_plan)
        self.autotune, self.tuned_rows32, self.tuned_waves16 = bool(autotune), None, None
        self.dev = batch["src"].device
        self.lib = _lib.load()
        cfgp = float(model.dropout.p) if p_drop is None else float(p_drop)
        self.p_drop = cfgp if model.training else 0.0
        self.seed = (int(seed) + ops.rank_seed_offset()) & 0x7FFFFFFFFFFFFFFF     # ranks draw different dropout masks
        self._validate(model, batch, labels=not self.module_mode)
        T, B = batch["src"].shape[0], batch["src"].shape[1]
        self.T, self.B = T, B
        self.shp = _lib.shape(B, T, model.d_inp, model.d_ob, d_pe=model.d_pe, nhead=model.nhead, nhid=model.nhid,
                              d_static=model.d_static if model.static else 0, n_classes=model.n_classes,
                              max_len=model.max_len)
        self.sp = ctypes.byref(self.shp)
        self.D = model.d_inp * model.d_ob + model.d_pe
        self.graph_info = model._graph(self.dev)                 # adjacency / ssum (built eagerly, once)
        self.ts = model.pos_encoder.timescales(self.dev)
        named = dict(model.named_parameters())
        gview = dict(zip(flat.names, flat.views))                # gradient slices in the flat buffer
        self.P = named
        self.G = gview
        self._alloc()
        self.seed_cell = torch.zeros(1, dtype=torch.int64, device=self.dev)
        # side branch for the trailing launches of the step (the weight-gradient reduces, the head's weight gradients: nothing in
        # the backward chain reads them).  MEASURED round 4, same box, 3 x 300 steps each: 0.554 ms/step with the branch against
        # 0.521 without -- the fourth time a forked graph loses here (the fork / join edges cost more than the 5-8 us launches
        # they take off the chain).  Off by default; RD_SIDE_REDUCE=1 turns it on (results are identical).
        self.side = torch.cuda.Stream(device=self.dev) if os.environ.get("RD_SIDE_REDUCE", "0") == "1" else None
        # trailing launches (the head's weight-gradient tiles, a layer's slice reduce) parked and appended to the next backward
        # chain launch as extra workgroups on its idle CUs (include/raindrop_hip.h rd_set_defer_trailing).  Also in the two-graph
        # data-parallel form: every part ends with rd_flush_trailing (_body), so the last layer's reduce -- whose results the first
        # gradient bucket's all-reduce needs between the graphs -- is launched on its own at the end of graph A instead of riding
        # in graph B; the head's tiles ride inside A, layer 0's reduce inside B.  RD_TRAILING_RIDE=0: every launch on its own (A/B).
        self.ride = self.side is None and os.environ.get("RD_TRAILING_RIDE", "1") != "0"
        # token plan: the step's fast paths only (fused message passing, row-block encoder, fused head)
        self.plan = None
        if self._want_plan and self.head_fused and self._plan_supported():
            self.plan = torch.zeros(max(int(self.lib.rd_token_plan_bytes(self.sp)) // 4, 64), dtype=torch.int32, device=self.dev)
        # all weight splits of the step in one launch (rd_step_prepare) where the shape takes prepared tiles
        enc_ok, k1_ok = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.call("rd_step_prepare_covers", self.sp, ctypes.byref(enc_ok), ctypes.byref(k1_ok))
        one = os.environ.get("RD_STEP_PREPARE", "1") != "0"
        self.one_begin = os.environ.get("RD_STEP_BEGIN", "1") != "0"      # A/B: token plan and weight splits as one launch
        self.prep_enc, self.prep_k1 = bool(enc_ok.value) and one, bool(k1_ok.value) and one
        self._prep_w = (ctypes.POINTER(_lib.RdEncoderPtrs) * self.nl)(*[ctypes.pointer(w) for w in self.enc_w])
        self._prep_saved = (ctypes.c_void_p * self.nl)(*[t.data_ptr() for t in self.enc_saved])
        self._prep_bytes = (ctypes.c_size_t * self.nl)(*[t.numel() for t in self.enc_saved])
        if self.split:
            self._check_split_order()
        self._ptrs = self._param_ptrs()                          # the captured graph / cached structs hold these addresses
        self.graph = None
        self.graph_b = None
        if use_graph:
            self._capture()

    @staticmethod
    def _validate(model, batch, labels=True):
        """The step hands raw data_ptr()s to the C-ABI: everything the autograd wrappers check per call is checked here
        once (dtype, contiguity, device, shapes, label range).  Labels are read on the host ONCE, at construction."""
        # The step enqueues the DEFAULT branch of the sensor stage (rd_sensor_stage_fwd / rd_msgpass_bwd: code/models_rd.py:317's
        # `use_beta = False`, distance exactly 0).  A model built with the paper's branch switched on would silently train a
        # different network here (and never see gradients for increase_dim / map_weights): refuse it.
        if getattr(model, "use_beta", False) or getattr(model, "compute_distance", False):
            raise _lib.RaindropHipError("TrainStep: Raindrop_v2(use_beta=True / compute_distance=True) runs on the eager model "
                                        "surface only (model.forward + autograd); the captured step implements the default branch")
        T, B = batch["src"].shape[0], batch["src"].shape[1]
        want = {"src": (torch.float32, (T, B, 2 * model.d_inp)), "times": (torch.float32, (T, B)),
                "lengths": (torch.int64, (B,))}
        if labels:
            want["y"] = (torch.int64, (B,))
        if model.static:
            want["static"] = (torch.float32, (B, model.d_static))
        dev = batch["src"].device
        for k, (dt, shape) in want.items():
            t = batch.get(k)
            if t is None or not t.is_cuda or t.device != dev:
                raise _lib.RaindropHipError("TrainStep: batch[%r] must be a tensor on %s" % (k, dev))
            if t.dtype != dt or tuple(t.shape) != shape or not t.is_contiguous():
                raise _lib.RaindropHipError("TrainStep: batch[%r] must be contiguous %s %s, got %s %s" % (
                    k, dt, shape, t.dtype, tuple(t.shape)))
        if T != model.max_len:
            raise _lib.RaindropHipError("TrainStep: src.shape[0] (%d) must equal max_len (%d)" % (T, model.max_len))
        if B > 0 and labels:
            lo, hi = int(batch["y"].min()), int(batch["y"].max())
            if lo < 0 or hi >= model.n_classes:
                raise _lib.RaindropHipError("TrainStep: labels must lie in [0, %d), got [%d, %d]" % (model.n_classes, lo, hi))

    @staticmethod
    def _validate_shapes_only(model, batch):
        """dtype / contiguity / device / shape / label checks of `_validate` without its refusal of the paper's branch (AutogradStep)"""
        ub, cd = getattr(model, "use_beta", False), getattr(model, "compute_distance", False)
        try:
            model.use_beta, model.compute_distance = False, False
            TrainStep._validate(model, batch)
        finally:
            model.use_beta, model.compute_distance = ub, cd

    def _param_ptrs(self):
        return tuple(p.data_ptr() for p in self.P.values()) + tuple(g.data_ptr() for g in self.G.values())

    def _plan_supported(self):
        m = self.model
        return plan_supported(m.d_inp, m.d_ob, self.T, self.D, m.nhead, m.nhid, self.lib.rd_get_precision())

    # ------------------------------------------------------------------------------------------
    def _alloc(self):
        lib, sp, dev, B, T, D = self.lib, self.sp, self.dev, self.B, self.T, self.D
        m = self.model
        f32 = dict(dtype=torch.float32, device=dev)
        # zero-filled: with a token plan parts of these buffers are never written, and a ghost product (x 0) of an uninitialised
        # NaN pattern would not be 0
        # Every buffer of the step is carved out of ONE allocation (round 6; RD_STEP_ARENA=0: separate torch allocations, A/B): the
        # large ones on 2-MB boundaries, the small ones packed into a common region.  Why: the K1 backward kernel's duration was
        # bimodal BETWEEN PROCESSES (16.4 vs 18.6 us, HISTORY round 5: "it follows how the process's memory is mapped") -- with one
        # contiguous mapping it is 16.2-16.6 us in every process and the step's kernel sum drops 0.6 % (four processes each,
        # alternating, one call: profiles/r06_step_arena_ab.txt).  Every kernel of a step starts on cold translations (~1 GB of
        # traffic since its last run); fewer, larger mappings are fewer walks.
        arena = os.environ.get("RD_STEP_ARENA", "1") == "1"
        if arena:
            nl_ = len(m.transformer_encoder.layers)
            al = lambda n: (max(int(n), 256) + (1 << 21) - 1) >> 21 << 21
            sizes = [al(T * B * D * 4)] * (1 + nl_ + 2) + [al(lib.rd_msgpass_saved_bytes(sp)), al(lib.rd_msgpass_workspace_bytes(sp))] + \
                    [al(lib.rd_encoder_layer_saved_bytes(sp))] * nl_ + [al(lib.rd_encoder_layer_workspace_bytes(sp))] * nl_
            small_cap = 16 << 20                                  # the small buffers' common region (head workspace, features, plan, ...)
            self._arena = torch.zeros(sum(sizes) + small_cap + (1 << 21), dtype=torch.uint8, device=dev)
            self._arena_off = (-self._arena.data_ptr()) % (1 << 21)
            self._small_off, self._small_end = self._arena_off, self._arena_off + small_cap
            self._arena_off += small_cap

            def carve(nbytes):
                n = max(int(nbytes), 256)
                if n < (1 << 20) and self._small_off + n <= self._small_end:
                    o = self._small_off
                    self._small_off += (n + 255) >> 8 << 8
                else:
                    o = self._arena_off
                    self._arena_off += al(n)
                    if self._arena_off > self._arena.numel():     # (sizes above are exact; a shape this list missed falls back)
                        return torch.zeros(n, dtype=torch.uint8, device=dev)
                return self._arena[o:o + n]
            u8 = carve
            zeros_f = lambda shape: carve(int(torch.Size(shape).numel()) * 4)[:int(torch.Size(shape).numel()) * 4].view(torch.float32).view(shape)
        else:
            u8 = lambda n: torch.zeros(max(int(n), 256), dtype=torch.uint8, device=dev)
            zeros_f = lambda shape: torch.zeros(shape, **f32)
        self._u8, self._zeros_f = u8, zeros_f
        self.z = zeros_f((T, B, D))
        self.mask = torch.empty((B, T), dtype=torch.bool, device=dev)
        self.k1_saved = u8(lib.rd_msgpass_saved_bytes(sp))
        self.k1_ws = u8(lib.rd_msgpass_workspace_bytes(sp))
        self.nl = len(m.transformer_encoder.layers)
        self.x = [self.z] + [zeros_f((T, B, D)) for _ in range(self.nl)]
        self.enc_saved = [u8(lib.rd_encoder_layer_saved_bytes(sp)) for _ in range(self.nl)]
        # one workspace per layer: a layer's trailing reduce launch (side branch, rd_set_side_stream) reads its partials while the
        # next layer's backward already writes its own
        self.enc_wss = [u8(lib.rd_encoder_layer_workspace_bytes(sp)) for _ in range(self.nl)]
        self.enc_ws = self.enc_wss[0]
        self.dx = [zeros_f((T, B, D)) for _ in range(2)]                   # ping-pong gradient buffers
        self.Fe = m.d_inp if m.static else 0
        self.feat = zeros_f((B, D + self.Fe))
        self.dfeat = zeros_f((B, D + self.Fe))
        self.hid = zeros_f((B, D + self.Fe))
        self.dhid = zeros_f((B, D + self.Fe))
        self.logits = zeros_f((B, m.n_classes))
        self.dlogits = zeros_f((B, m.n_classes))
        self.loss = torch.zeros((), **f32)
        dh = D + self.Fe
        self.wg_ws = u8(max(lib.rd_linear_bwd_weight_workspace_bytes(B, dh, dh),
                            lib.rd_linear_bwd_weight_workspace_bytes(B, m.n_classes, dh),
                            lib.rd_linear_bwd_weight_workspace_bytes(B, max(self.Fe, 1), max(m.d_static, 1))))
        # classifier head + loss + their backward as two launches (rd_head.hip) when the sizes fit, else operator by operator
        self.head_fused = bool(lib.rd_head_train_supported(D, self.Fe, m.n_classes))
        self.head_ws = u8(lib.rd_head_train_workspace_bytes(B, dh, m.n_classes)) if self.head_fused else None
        self.enc_w = []
        self.enc_g = []
        for i, layer in enumerate(m.transformer_encoder.layers):
            pre = "transformer_encoder.layers.%d." % i
            self.enc_w.append(_lib.RdEncoderPtrs(*[self.P[pre + n].data_ptr() for n in ops.ENC_PARAM_NAMES]))
            self.enc_g.append(_lib.RdEncoderPtrs(*[self.G[pre + n].data_ptr() for n in ops.ENC_PARAM_NAMES]))

    # ------------------------------------------------------------------------------------------
    def _call(self, name, *a):
        _lib.call(name, *a)

    def _body(self, part=None):
        """_body_impl + the join of the side branch: launches forked inside this part (weight-gradient reduces, the head's weight
        gradients: rd_set_side_stream) are complete, in stream order, when the part is."""
        self._body_impl(part)
        _lib.call("rd_flush_trailing", ops._stream())              # a parked trailing launch nobody picked up (layer 0's reduce)
        _lib.call("rd_side_join", ops._stream())

    def _body_impl(self, part=None):
        """Enqueue one forward + loss + backward on the current stream (no host sync).  part 'a' / 'b': the two halves of the split
        form (see __init__); 'begin' / 'k1f' / 'mid' / 'k1b': the step cut around the message-passing stage, 'enc' / 'head' / 'encb':
        'mid' cut further into encoder forward | head + loss | encoder backward (capture_segments: bench.py times the K1 launches
        and the encoder layers as they run INSIDE the step); None: everything."""
        if part == "b":
            return self._body_tail(self.nl - 2)
        m, b, P, G, sp = self.model, self.batch, self.P, self.G, self.sp
        st = ops._stream()
        B, T, D, Fe = self.B, self.T, self.D, self.Fe
        dh = D + Fe
        c = self._call
        W1, b1 = P["ob_propagation.lin_value.weight"], P["ob_propagation.lin_value.bias"]
        W2, b2 = P["ob_propagation_layer2.lin_value.weight"], P["ob_propagation_layer2.lin_value.bias"]
        ssum = self.graph_info["ssum"]
        if part == "k1b":
            return self._k1_bwd(self.dx[self.nl % 2], st)
        if part == "mb":                                  # module mode: backward from the caller's d loss / d logits
            self._head_module(self.dx[0], st, backward=True)
            return self._body_tail(self.nl - 1, self.dx[0])
        if part in (None, "a", "begin", "mf"):
            prep = self.prep_enc or self.prep_k1
            prep_args = (self.nl if self.prep_enc else 0, self._prep_w, self._prep_saved, self._prep_bytes,
                         _p(W1) if self.prep_k1 else None, _p(W2) if self.prep_k1 else None, _p(self.k1_saved), self.k1_saved.numel(), st)
            cell = _p(self.seed_cell) if self.p_drop > 0.0 else None
            if self.plan is not None and prep and self.one_begin:             # plan + seed bump + every weight split: ONE launch
                c("rd_step_begin", sp, _p(b["lengths"]), _p(self.plan), cell, 1, *prep_args)
            else:
                if self.plan is not None:                                      # lengths -> token plan (+ the seed bump: one launch)
                    c("rd_token_plan", sp, _p(b["lengths"]), _p(self.plan), cell, 1, st)
                elif self.p_drop > 0.0:
                    c("rd_seed_cell_advance", _p(self.seed_cell), 1, st)       # fresh masks per replay
                if prep:
                    c("rd_step_prepare", sp, *prep_args)
            if part == "begin":
                return
        # ---------------- forward ----------------
        if part in (None, "a", "k1f", "mf"):
            c("rd_sensor_stage_fwd_prepared" if self.prep_k1 else "rd_sensor_stage_fwd", sp, _p(b["src"]), _p(b["times"]), _p(b["lengths"]), _p(self.ts), _p(P["R_u"]), _p(W1),
              _p(b1), _p(W2), _p(b2), _p(ssum), self.p_drop, self.seed, _p(self.z), _p(self.mask), _p(self.k1_saved),
              self.k1_saved.numel(), st)
            if part == "k1f":
                return
        if part == "encb":
            return self._body_tail(self.nl - 1, self.dx[0], k1=False)
        if part != "head":
            for i in range(self.nl):
                c("rd_encoder_layer_fwd", sp, i | (0x10000 if self.prep_enc else 0), _p(self.x[i]), _p(self.mask), ctypes.byref(self.enc_w[i]), self.p_drop,
                  self.seed, _p(self.x[i + 1]), _p(self.enc_saved[i]), self.enc_saved[i].numel(), _p(self.enc_wss[i]),
                  self.enc_wss[i].numel(), st)
            if part == "enc":
                return
        cur = self.dx[0]
        if part == "mf":                                  # module mode: forward ends with the logits
            return self._head_module(cur, st, backward=False)
        if self.head_fused:
            e = (lambda n: _p(P[n]) if Fe else None)
            ge = (lambda n: _p(G[n]) if Fe else None)
            c("rd_head_train", sp, D, m.d_static if Fe else 0, Fe, m.n_classes, _p(self.x[-1]), _p(self.mask), _p(b["lengths"]),
              _p(b["static"]) if Fe else None, e("emb.weight"), e("emb.bias"), _p(P["mlp_static.0.weight"]),
              _p(P["mlp_static.0.bias"]), _p(P["mlp_static.2.weight"]), _p(P["mlp_static.2.bias"]), _p(b["y"]), _p(self.loss),
              _p(self.logits), ge("emb.weight"), ge("emb.bias"), _p(G["mlp_static.0.weight"]), _p(G["mlp_static.0.bias"]),
              _p(G["mlp_static.2.weight"]), _p(G["mlp_static.2.bias"]), _p(cur), _p(self.head_ws), self.head_ws.numel(), st)
        else:
            self._head_by_operator(cur, st)
        if part == "head":
            return
        if part == "a":                                   # the last layer's backward closes part A
            self._enc_bwd(self.nl - 1, cur, self.dx[1], st)
            return
        self._body_tail(self.nl - 1, cur, k1=(part != "mid"))

    def _enc_bwd(self, i, cur, nxt, st):
        self._call("rd_encoder_layer_bwd", self.sp, i, _p(self.x[i]), _p(self.mask), ctypes.byref(self.enc_w[i]), self.p_drop,
                   self.seed, _p(self.enc_saved[i]), self.enc_saved[i].numel(), _p(cur), _p(nxt), ctypes.byref(self.enc_g[i]),
                   _p(self.enc_wss[i]), self.enc_wss[i].numel(), st)

    def _body_tail(self, top, cur=None, k1=True):
        """Backward of encoder layers top .. 0 and (k1) of the sensor stage; the entry gradient is dx[0] for the top layer of the
        stack (written by the head) and alternates between the two buffers from there."""
        st = ops._stream()
        if cur is None:                                   # layer `top` reads what layer top + 1 wrote
            cur = self.dx[(self.nl - 1 - top) % 2]
        for i in range(top, -1, -1):
            nxt = self.dx[1] if cur is self.dx[0] else self.dx[0]
            self._enc_bwd(i, cur, nxt, st)
            cur = nxt
        if k1:
            self._k1_bwd(cur, st)

    def _k1_bwd(self, cur, st):
        b, P, G, sp = self.batch, self.P, self.G, self.sp
        W1, W2 = P["ob_propagation.lin_value.weight"], P["ob_propagation_layer2.lin_value.weight"]
        self._call("rd_msgpass_bwd", sp, _p(b["src"]), _p(P["R_u"]), _p(W1), _p(W2), _p(self.graph_info["ssum"]), self.p_drop,
                   _p(self.k1_saved), self.k1_saved.numel(), _p(self.z), _p(cur), self.D, _p(G["ob_propagation.lin_value.weight"]),
                   _p(G["ob_propagation.lin_value.bias"]), _p(G["ob_propagation_layer2.lin_value.weight"]),
                   _p(G["ob_propagation_layer2.lin_value.bias"]), _p(G["R_u"]), _p(self.k1_ws), self.k1_ws.numel(), st)

    def _head_module(self, cur, st, backward):
        """The classifier head around a loss the caller evaluates (module mode): rd_head_forward -> self.logits, rd_head_backward
        from self.dlogits -> the head's parameter gradients and the encoder stack's entry gradient `cur`."""
        m, b, P, G, sp = self.model, self.batch, self.P, self.G, self.sp
        D, Fe = self.D, self.Fe
        e = (lambda n: _p(P[n]) if Fe else None)
        ge = (lambda n: _p(G[n]) if Fe else None)
        common = (sp, D, m.d_static if Fe else 0, Fe, m.n_classes, _p(self.x[-1]), _p(self.mask), _p(b["lengths"]),
                  _p(b["static"]) if Fe else None, e("emb.weight"), e("emb.bias"), _p(P["mlp_static.0.weight"]),
                  _p(P["mlp_static.0.bias"]), _p(P["mlp_static.2.weight"]), _p(P["mlp_static.2.bias"]))
        if not backward:
            return self._call("rd_head_forward", *common, _p(self.logits), _p(self.head_ws), self.head_ws.numel(), st)
        self._call("rd_head_backward", *common, _p(self.dlogits), ge("emb.weight"), ge("emb.bias"), _p(G["mlp_static.0.weight"]),
                   _p(G["mlp_static.0.bias"]), _p(G["mlp_static.2.weight"]), _p(G["mlp_static.2.bias"]), _p(cur), _p(self.head_ws),
                   self.head_ws.numel(), st)

    def _head_by_operator(self, cur, st):
        """masked mean -> [agg | emb] -> mlp_static -> cross entropy and their backward, one C-ABI call per operator."""
        m, b, P, G, sp = self.model, self.batch, self.P, self.G, self.sp
        B, D, Fe = self.B, self.D, self.Fe
        dh = D + Fe
        c = self._call
        c("rd_masked_mean_fwd", sp, D, _p(self.x[-1]), _p(self.mask), _p(b["lengths"]), _p(self.feat), dh, st)
        if Fe:
            emb_out = self.feat[:, D:]                                       # right block of [agg | emb]
            c("rd_linear_fwd", B, Fe, m.d_static, _p(b["static"]), m.d_static, _p(P["emb.weight"]), _p(P["emb.bias"]),
              ctypes.c_void_p(emb_out.data_ptr()), dh, 0, st)
        c("rd_linear_fwd", B, dh, dh, _p(self.feat), dh, _p(P["mlp_static.0.weight"]), _p(P["mlp_static.0.bias"]),
          _p(self.hid), dh, 1, st)
        C = m.n_classes
        c("rd_linear_fwd", B, C, dh, _p(self.hid), dh, _p(P["mlp_static.2.weight"]), _p(P["mlp_static.2.bias"]),
          _p(self.logits), C, 0, st)
        # ---------------- loss: mean cross entropy (code/Raindrop.py:255,322) and its gradient ----------
        c("rd_softmax_xent", B, C, _p(self.logits), _p(b["y"]), _p(self.loss), _p(self.dlogits), st)
        # ---------------- backward ----------------
        ws, wsn = _p(self.wg_ws), self.wg_ws.numel()
        c("rd_linear_bwd_weight", B, C, dh, _p(self.dlogits), C, _p(self.hid), dh, _p(G["mlp_static.2.weight"]),
          _p(G["mlp_static.2.bias"]), ws, wsn, st)
        c("rd_linear_bwd_input_gated", B, C, dh, _p(self.dlogits), C, _p(P["mlp_static.2.weight"]), _p(self.hid), dh,
          _p(self.dhid), dh, st)                                             # ReLU gate of mlp_static[1] folded in
        c("rd_linear_bwd_weight", B, dh, dh, _p(self.dhid), dh, _p(self.feat), dh, _p(G["mlp_static.0.weight"]),
          _p(G["mlp_static.0.bias"]), ws, wsn, st)
        c("rd_linear_bwd_input", B, dh, dh, _p(self.dhid), dh, _p(P["mlp_static.0.weight"]), _p(self.dfeat), dh, st)
        if Fe:
            demb = self.dfeat[:, D:]
            c("rd_linear_bwd_weight", B, Fe, m.d_static, ctypes.c_void_p(demb.data_ptr()), dh, _p(b["static"]),
              m.d_static, _p(G["emb.weight"]), _p(G["emb.bias"]), ws, wsn, st)
        c("rd_masked_mean_bwd", sp, D, _p(self.dfeat), dh, _p(self.mask), _p(b["lengths"]), _p(cur), st)

    def _with_cell(self, fn):
        """Run `fn` with this step's seed cell registered.  The registration is read when a kernel is ENQUEUED (the pointer
        travels as a kernel argument), so it is scoped to the enqueue / the capture: a captured graph keeps the cell it was
        captured with, and nothing else in the process (an eager model, another TrainStep) ever sees this step's cell --
        dropping a TrainStep can no longer leave a dangling pointer behind."""
        _lib.call("rd_set_seed_cell", _p(self.seed_cell))
        _lib.call("rd_set_token_plan", _p(self.plan))
        _lib.call("rd_set_side_stream", ctypes.c_void_p(self.side.cuda_stream) if self.side is not None else None)
        _lib.call("rd_set_defer_trailing", 1 if self.ride else 0)
        try:
            return fn()
        finally:
            _lib.call("rd_set_seed_cell", None)
            _lib.call("rd_set_token_plan", None)
            _lib.call("rd_set_side_stream", None)
            _lib.call("rd_set_defer_trailing", 0)

    def _capture(self):
        """Capture the step as one hipGraph (two in the split form).  With `autotune`, the step is captured once per setting of the
        library's tuning knobs (32-row vs 64-row workgroups of the encoder's row-block products, rd_set_rowgemm_rows32, then 8 vs
        16 waves for the plain ones: which is faster depends on the device, 8 % either way was measured on two boxes of one pool)
        and the fastest kept.  Under torch.distributed every rank times every variant and the per-variant times are SUMMED over
        the ranks before the choice, so all ranks run the same kernels -- and the same ones a single process would pick on such
        boxes.  The choice is in `tuned_rows32` / `tuned_waves16` (bench.py: config.tuned).
        Results: the knobs change which rows share a workgroup, never a row's arithmetic; on the fused row-local chains
        (rd_encfuse.hip: the P19 / P12 widths) no cross-row sum depends on them, so every variant gives the same gradient bits.
        On the unfused LayerNorm-epilogue products (other widths) the dgamma / dbeta partial grouping follows the workgroup
        height: there the bits depend on the choice, which is why it is recorded (pin it with RD_RG_ROWS32 / RD_RG_WAVES16)."""
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if not self.autotune or os.environ.get("RD_RG_ROWS32") is not None or os.environ.get("RD_RG_WAVES16") is not None:
            return self._capture_one()

        def agree(ts):
            """per-variant times summed over the ranks (identical on every rank afterwards)"""
            if not multi:
                return ts
            dev = self.dev if dist.get_backend() == "nccl" else torch.device("cpu")
            t = torch.tensor(ts, dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return [float(v) for v in t.cpu()]

        def timed(r32, w16):
            _lib.call("rd_set_rowgemm_rows32", r32)
            _lib.call("rd_set_rowgemm_waves16", w16)
            self._capture_one()
            graphs = (self.graph, self.graph_b)

            def replay():
                graphs[0].replay()
                if graphs[1] is not None:
                    graphs[1].replay()
            for _ in range(3):
                replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                replay()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0, graphs)
        # workgroup height first (all / none / plain products only / LayerNorm-fused ones only) at the default wave counts,
        # then the wave count of the plain products at the best height
        heights = (15, 0, 3, 12)
        runs = [timed(r32, 12) for r32 in heights]
        ts = agree([r[0] for r in runs])
        bi = min(range(len(heights)), key=lambda i: (ts[i], i))
        best_r32, best_w16, best_graphs, best_t = heights[bi], 12, runs[bi][1], ts[bi]
        alt = timed(best_r32, 15)
        alt_t = agree([alt[0]])[0]
        if alt_t < best_t:
            best_w16, best_graphs = 15, alt[1]
        self.graph, self.graph_b = best_graphs
        self.tuned_rows32, self.tuned_waves16 = best_r32, best_w16
        _lib.call("rd_set_rowgemm_rows32", best_r32)                 # eager calls of this process follow the same choice
        _lib.call("rd_set_rowgemm_waves16", best_w16)

    def _capture_one(self):
        def cap():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):                                         # warm-up: lazy inits happen here
                    self._body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            self.graph_b = None
            if not self.split:
                with torch.no_grad(), torch.cuda.graph(self.graph):
                    self._body()
                return
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self._body("a")
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph_b, pool=self.graph.pool()):
                self._body("b")
        self._with_cell(cap)

    def capture_segments(self, parts=("begin", "k1f", "mid", "k1b")):
        """The same step as consecutive hipGraphs, one per part (measurement only: bench.py brackets the 'k1f' / 'k1b' replays with
        HIP events, so the message-passing launches are timed with the cache state, clocks and neighbours they have in the step).
        Replaying the graphs in order is one step."""
        graphs = []

        def cap():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):
                    for pt in parts:
                        self._body(pt)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            pool = None
            for pt in parts:
                g = torch.cuda.CUDAGraph()
                with torch.no_grad(), (torch.cuda.graph(g) if pool is None else torch.cuda.graph(g, pool=pool)):
                    self._body(pt)
                pool = g.pool()
                graphs.append(g)
        self._with_cell(cap)
        return graphs

    def capture_marked(self, parts=("begin", "k1f", "enc", "head", "encb", "k1b")):
        """The step as ONE hipGraph with an external timing event recorded in front of every part and behind the last
        (torch.cuda.Event(enable_timing=True, external=True): event-record NODES of the graph): per-part device times of the real
        step, without the ~10 us a graph boundary costs per segment.  Returns (graph, events); raises where the runtime cannot
        capture external event records (callers fall back to capture_segments)."""
        events = [torch.cuda.Event(enable_timing=True, external=True) for _ in range(len(parts) + 1)]
        out = []

        def cap():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):
                    for pt in parts:
                        self._body(pt)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g):
                for k, pt in enumerate(parts):
                    events[k].record()
                    self._body(pt)
                events[len(parts)].record()
            out.append(g)
        self._with_cell(cap)
        return out[0], events

    # ------------------------------------------------------------------------------------------
    def _early_names(self):
        """Parameters whose gradients the first all-reduce bucket of the split form carries: the last encoder layer's and the
        classifier head's (mlp_static).  (The static embedding's gradient is complete by then too, but sits in front of the
        offset in forward order and travels with the rest: early completion is harmless, late completion is not.)"""
        pre = "transformer_encoder.layers.%d." % (self.nl - 1)
        return [n for n in self.flat.names if n.startswith(pre) or n.startswith("mlp_static.")]

    def _check_split_order(self):
        """Ordering contract of the split (overlapped all-reduce) form: in the flat buffer every gradient the first graph completes
        must sit at or behind early_grad_offset(), and nothing the SECOND graph writes (R_u, ob_propagation*, earlier encoder
        layers, emb) may sit there -- the collective started between the graphs would read it while graph B is still writing.
        raindrop_amd.synth.live_parameter_names (forward order) satisfies it; model.named_parameters() order does NOT
        (R_u and ob_propagation* are registered behind transformer_encoder, code/models_rd.py:241-247)."""
        names = list(self.flat.names)
        first = "transformer_encoder.layers.%d.self_attn.in_proj_weight" % (self.nl - 1)
        if first not in names:
            raise _lib.RaindropHipError("TrainStep(split=True): %s is not in the flat gradient buffer" % first)
        i0 = names.index(first)
        early = set(self._early_names())
        bad_tail = [n for n in names[i0:] if n not in early]
        bad_head = [n for n in names[:i0] if n in early]
        if bad_tail or bad_head:
            raise _lib.RaindropHipError(
                "TrainStep(split=True): the flat gradient buffer must hold the parameters in FORWARD order "
                "(raindrop_amd.synth.live_parameter_names): behind %s only the last encoder layer and mlp_static may follow; "
                "found %s behind it and %s in front of it.  Pass split=False (or RD_DP_OVERLAP=0) for another order."
                % (first, bad_tail[:4], bad_head[:4]))

    def early_grad_offset(self):
        """Offset in the flat gradient buffer from which every gradient is final when `between` runs (split form): the last encoder
        layer's parameters and, behind them in forward order, the classifier head's."""
        return self.flat.tail_start("transformer_encoder.layers.%d.self_attn.in_proj_weight" % (self.nl - 1))

    def run(self, between=None):
        """One forward + loss + backward; gradients land in flat.flat (p.grad views point there).  `between` (split form only) is
        called after the first graph has been enqueued and before the second: gradients at flat offsets >= early_grad_offset() are
        complete in stream order at that point."""
        if self._param_ptrs() != self._ptrs:
            raise _lib.RaindropHipError("TrainStep: a parameter or gradient buffer moved since construction (model.to(), "
                                        "flatten_parameters() or a re-assignment): build a new TrainStep")
        if self.graph is not None:
            self.graph.replay()
            if self.graph_b is not None:
                if between is not None:
                    between()
                self.graph_b.replay()
        else:
            def eager():
                with torch.no_grad():
                    if self.split:
                        self._body("a")
                        if between is not None:
                            between()
                        self._body("b")
                    else:
                        self._body()
            self._with_cell(eager)
        for p, v in zip(self.flat.params, self.flat.views):
            p.grad = v
        return self.loss

    def run_allreduce(self):
        """run() + the gradient all-reduce of `flat` (no-op for one process).  Split form: the collective of the gradients the first
        graph completes (last encoder layer + head: ~0.8 of 2.0 MB at P19) is started between the two graphs and runs beside the
        rest of the backward pass; the remainder follows the second graph.  Returns the loss tensor."""
        flat = self.flat
        if not self.split:
            loss = self.run()
            flat.allreduce()
            return loss
        off, hold = self.early_grad_offset(), []
        loss = self.run(between=lambda: hold.append(flat.allreduce_range_async(off, flat.flat.numel())))
        rest = flat.allreduce_range_async(0, off)
        for h in hold:
            flat.allreduce_wait(h)
        flat.allreduce_wait(rest)
        return loss

    # ------------------------------------------------------------------------------------------
    def capture_full(self, opt):
        """The WHOLE step as ONE hipGraph (round 5; opt-in): forward + loss + backward, the gradient all-reduce(s) and the optimizer.
        At N > 1 the first bucket's collective (last encoder layer + head) is started between the two halves of the backward and
        joined behind the second one's -- the collectives' own stream forks from and joins the captured stream (RCCL supports
        stream capture; tested on a one-rank group: tests/test_dp_gpu.py) --, then `opt.step_captured()` (FlatAdam: device step
        cell + rd_adam_step_dev).  The host side of a step is then ONE replay (run_full).  Raises whatever the capture raises;
        the caller falls back to run_allreduce() + opt.step()."""
        flat = self.flat
        opt.sync_step_cell()

        def cap():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(2):                                         # warm-up outside the capture (RCCL's lazy inits, too)
                    self._body()
                    flat.allreduce()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # the optimizer's device step state is advanced by the step's FIRST launch (rd_step_begin, next to the seed bump) where
            # the step has that launch -- registered for the capture only --, else by a one-thread launch in front of the update
            begin_adv = self.plan is not None and (self.prep_enc or self.prep_k1) and self.one_begin
            try:
                with torch.no_grad(), torch.cuda.graph(g):
                    if begin_adv:
                        opt.register_cell(True)
                    if self.split:
                        off = self.early_grad_offset()
                        self._body("a")
                        opt.register_cell(False)
                        h1 = flat.allreduce_range_async(off, flat.flat.numel())
                        self._body("b")
                        h0 = flat.allreduce_range_async(0, off)
                        flat.allreduce_wait(h1)
                        flat.allreduce_wait(h0)
                    else:
                        self._body()
                        opt.register_cell(False)
                        flat.allreduce()
                    opt.step_captured(advance=not begin_adv)
            finally:
                opt.register_cell(False)
            self.graph_full = g
        self._with_cell(cap)
        opt.sync_step_cell()                                               # the warm-up did not step the optimizer; neither did the capture
        self._full_opt = opt
        self._full_hyper = opt.hyper()                                     # betas, eps are launch arguments: baked into the graph (lr, weight decay: device cell)
        return self.graph_full

    def run_full(self):
        """One replay of capture_full's graph = one training step including the optimizer; returns the loss tensor."""
        if self._param_ptrs() != self._ptrs:
            raise _lib.RaindropHipError("TrainStep: a parameter or gradient buffer moved since construction: build a new TrainStep")
        if self._full_opt.hyper() != self._full_hyper:                     # betas / eps changed: launch constants of the captured
            cell = self.seed_cell.clone()                                  # Adam -- capture again
            self.capture_full(self._full_opt)                              # (the warm-up passes advance the dropout stream: put it back)
            self.seed_cell.copy_(cell)
        self._full_opt.sync_cell_hyper()                                   # lr / weight decay (ReduceLROnPlateau, code/Raindrop.py:257-259;
                                                                           # warm-up / cosine schedules): an 8-byte copy, no new capture
        if getattr(self._full_opt, "_cell_stale", False):                  # host-side steps were taken since: device state follows self.t
            self._full_opt.sync_step_cell()
            self._full_opt._cell_stale = False
        self.graph_full.replay()
        self._full_opt.note_replay()
        for p, v in zip(self.flat.params, self.flat.views):
            p.grad = v
        return self.loss

    def close(self):
        """Kept for callers of the round-1 API: the seed cell is no longer registered outside run() / capture."""
        self.graph_full = None
        self.graph = None
        self.graph_b = None


class AutogradStep:
    """Any `Raindrop_v2` -- the paper's branch (`use_beta=True` / `compute_distance=True`), which `TrainStep` refuses, included -- as ONE
    hipGraph per training step: `model.forward -> CrossEntropyLoss -> loss.backward() -> Adam` exactly as code/Raindrop.py:319-324
    runs them, through the module's own autograd surface (one C-ABI call per operator), captured once with static input buffers and
    replayed.  Same kernels, same results as the eager loop (the capture only removes the host from the replay path); dropout masks
    change per replay through a device seed cell the graph bumps itself, as in `TrainStep`.

    This is the composed form of the use_beta path (obs embed -> lin_value / increase_dim -> LDS graph operator -> per-sample edge
    softmax -> layer 2 -> tokens: ~7 launches for the sensor stage, DESIGN.md (e')), not a fused kernel; `bench.py --use-beta`
    times it.  The optimizer is torch's own Adam in its capturable fused form over the model's parameters (the reference's
    `torch.optim.Adam(model.parameters(), lr)`, code/Raindrop.py:256), inside the graph; `lr` is a device tensor there, so a
    scheduler can change it without a new capture."""

    def __init__(self, model, batch, lr=1e-4, optimizer=True, seed=1234):
        self.model, self.batch = model, batch
        self.dev = batch["src"].device
        TrainStep._validate_shapes_only(model, batch)
        self.seed_cell = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.opt = None
        if optimizer:
            self.opt = torch.optim.Adam(self.params, lr=torch.tensor(float(lr), device=self.dev), capturable=True, fused=True)
        self.loss = torch.zeros((), dtype=torch.float32, device=self.dev)
        self.logits = None
        self.distance = None
        model.graph_step = False                                  # the operator surface is what gets captured
        self._capture()

    def _one(self):
        b = self.batch
        logits, distance, _ = self.model(b["src"], b["static"], b["times"], b["lengths"])
        loss = torch.nn.functional.cross_entropy(logits, b["y"])
        loss.backward()
        if self.opt is not None:
            self.opt.step()
        return logits, distance, loss

    def _capture(self):
        _lib.call("rd_set_seed_cell", _p(self.seed_cell))
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):                                 # warm-up: lazy initialisations, the optimizer's state tensors
                    for p in self.params:
                        p.grad = None
                    self._one()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for p in self.params:
                p.grad = None                                      # the captured backward ALLOCATES the gradients (static addresses)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                _lib.call("rd_seed_cell_advance", _p(self.seed_cell), 1, ops._stream())
                logits, distance, loss = self._one()
                self.loss_static, self.logits, self.distance = loss.detach(), logits.detach(), distance.detach()
        finally:
            _lib.call("rd_set_seed_cell", None)

    def run(self):
        self.graph.replay()
        return self.loss_static

    def close(self):
        self.graph = None
