"""The captured step behind the nn.Module surface: what an UNCHANGED training loop gets.

code/Raindrop.py:310-324 runs, per batch,

    outputs, local_structure_regularization, _ = model.forward(P, Pstatic, Ptime, lengths)
    loss = criterion(outputs, y);  loss.backward();  optimizer.step()

By DEFAULT (round 5; `RD_MODULE_GRAPH=0` or `model.graph_step = False` switch it off) `Raindrop_v2.forward` routes a TRAINING call
through two hipGraphs instead of one C-ABI call per operator under autograd:

  * forward  = the inputs copied into the step's static buffers, then graph F: token plan + weight splits, sensor stage,
               encoder layers, classifier head up to the logits (`TrainStep` part 'mf');
  * backward = one autograd node for the whole model: the loop's d loss / d logits copied in, then graph B: head backward,
               encoder backward, sensor-stage backward (part 'mb') -- every live parameter's gradient lands in one flat buffer.
               Round 6: each `p.grad` is then SET to a persistent view of that buffer by the node itself (35 attribute stores)
               instead of handing autograd 35 freshly made views of a 2-MB clone (70 tensor constructions = ~0.14 ms of host time
               per step, most of what the drop-in loop cost over the explicit step).  `optimizer.zero_grad()` (either form) and
               torch's own Adam behave as before; a second backward without zero_grad in between ACCUMULATES as autograd would
               (the node notices that p.grad still is its own view and adds the previous gradients back in);
               `RD_MODULE_GRAD_VIEWS=0` restores the clone-and-return form (parameter hooks then fire as with the eager path).

Same kernels, token plan and dropout scheme as `raindrop_amd.step.TrainStep` (masks are a function of the step's seed cell, which
the forward graph bumps per replay); the loss stays the caller's.  Calls the captured step does not cover fall back to the eager
operators, silently and per call: evaluation / no-grad calls, `use_beta` / `compute_distance` models, another batch size or
device than the captured one is handled by capturing a second runner (the `RD_MODULE_GRAPH_MAX` = 4 most recently created are
kept: each holds a step's activations), torch.distributed with more than one rank (use `TrainStep` + `dp.FlatGradAllReduce`
there).  A capture that fails -- sizes outside the fused head, or an error inside the capture -- is reported once per shape with a
warning, and those calls run on the eager operators.

One captured forward may be outstanding at a time (the backward graph reads the activations the forward graph left): a training
forward that arrives while the previous captured call's autograd node is still alive and has not run its backward takes the eager
path for that call (both backwards then work, as with the eager operators); should a stale node's backward be reached anyway it
raises instead of returning wrong gradients."""
import os
import weakref

import torch

from . import _lib, dp, synth


def enabled(model):
    flag = getattr(model, "graph_step", None)
    if flag is None:
        flag = os.environ.get("RD_MODULE_GRAPH", "1") != "0"
    return bool(flag)


def _max_runners():
    try:
        return max(1, int(os.environ.get("RD_MODULE_GRAPH_MAX", "4")))
    except ValueError:
        return 4


class _Runner:
    """One captured (forward graph, backward graph) pair for one model, batch shape and device."""

    def __init__(self, model, T, B, dev):
        from .step import TrainStep
        nl = len(model.transformer_encoder.layers)
        names = synth.live_parameter_names(dict(static=model.static, nlayers=nl))
        named = dict(model.named_parameters())
        self.names = names
        self.params = [named[n] for n in names]
        old_grads = [p.grad for p in self.params]
        self.flat = dp.FlatGradAllReduce(list(zip(names, self.params)), n_buckets=1)
        for p, g in zip(self.params, old_grads):                       # the flat buffer is the step's output here, not p.grad's home
            p.grad = g
        f32 = dict(dtype=torch.float32, device=dev)
        self.batch = dict(src=torch.zeros((T, B, 2 * model.d_inp), **f32), times=torch.zeros((T, B), **f32),
                          lengths=torch.zeros((B,), dtype=torch.int64, device=dev),
                          static=torch.zeros((B, model.d_static), **f32) if model.static else None)
        self.step = TrainStep(model, self.flat, self.batch, use_graph=False, autotune=False, split=False, module_mode=True)
        if not self.step.head_fused:
            raise _lib.RaindropHipError("graph_module: classifier head sizes outside rd_head_forward / rd_head_backward")
        self.graph_f, self.graph_b = self.step.capture_segments(("mf", "mb"))
        self.ptrs = self._ptrs()
        self.assign = os.environ.get("RD_MODULE_GRAD_VIEWS", "1") != "0"
        self.fdst = [self.batch["src"], self.batch["times"]] + ([self.batch["static"]] if model.static else [])
        self.gen = 0
        self.last_ctx = None                                           # weak reference to the autograd node of the latest captured forward

    def busy(self):
        """the latest captured forward still waits for its backward (its autograd node is alive and has not run)"""
        c = self.last_ctx() if self.last_ctx is not None else None
        return c is not None and not getattr(c, "rd_done", True)

    def _ptrs(self):
        return tuple(p.data_ptr() for p in self.params)

    def stale(self):
        """parameters were moved or replaced since the capture (the graphs hold their addresses)"""
        return self._ptrs() != self.ptrs

    def forward(self, src, static, times, lengths):
        b = self.batch
        torch._foreach_copy_(self.fdst, [src, times, static] if b["static"] is not None else [src, times])   # one launch
        b["lengths"].copy_(lengths)
        self.gen += 1
        self.graph_f.replay()
        return self.step.logits.clone()

    def backward(self, dlogits, needs=None):
        self.step.dlogits.copy_(dlogits)
        if not self.assign:
            self.graph_b.replay()
            g = self.flat.flat.clone()                                 # a fresh buffer per step: autograd may keep (or steal) the views
            return [g[o:e].view_as(p) for (o, e), p in zip(self.flat.slices, self.params)]
        # p.grad <- persistent views of the step's flat gradient buffer, set here (nothing is returned to autograd for the parameters).
        # A parameter whose .grad still IS our view was not zeroed since the last backward: the caller accumulates -- keep the old
        # gradients aside and add them back (the replay overwrites the buffer the views alias).
        views = self.flat.views
        keep = None
        acc = [p.grad is v for p, v in zip(self.params, views)]
        if any(acc):
            keep = self.flat.flat.clone()
            if not all(acc):                                           # slices that are not part of the accumulation must not be added back
                for a_, (o, e) in zip(acc, self.flat.slices):
                    if not a_:
                        keep[o:e].zero_()
        self.graph_b.replay()
        if keep is not None:
            self.flat.flat.add_(keep)
        for i, (p, v) in enumerate(zip(self.params, views)):
            if needs is not None and not needs[i]:
                continue
            if p.grad is None or p.grad is v:
                p.grad = v
            else:                                                      # a foreign gradient is there already (another loss term): accumulate into it
                p.grad = p.grad + v
        return None


class _GraphStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, src, static, times, lengths, *params):
        ctx.runner = runner
        out = runner.forward(src, static, times, lengths)
        ctx.gen = runner.gen
        ctx.needs = [p.requires_grad for p in params]
        ctx.rd_done = False
        runner.last_ctx = weakref.ref(ctx)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dlogits):
        r = ctx.runner
        if ctx.gen != r.gen:
            raise _lib.RaindropHipError("graph_module: backward of a forward call that is no longer the latest one -- the captured "
                                        "step keeps ONE set of activations (run forward and backward in pairs, or unset "
                                        "RD_MODULE_GRAPH / model.graph_step for this pattern)")
        grads = r.backward(dlogits.contiguous().float(), ctx.needs)
        ctx.rd_done = True
        if grads is None:
            return (None, None, None, None, None) + (None,) * len(ctx.needs)
        return (None, None, None, None, None) + tuple(g if need else None for g, need in zip(grads, ctx.needs))


def forward(model, src, static, times, lengths):
    """logits [B,C] through the captured step, or None where this call is not covered (the caller continues on the eager path)."""
    if not (model.training and torch.is_grad_enabled()) or model.use_beta or model.compute_distance:
        return None
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return None
    if model.static and static is None:
        return None
    T, B = src.shape[0], src.shape[1]
    if B == 0 or src.device.type != "cuda":
        return None
    dev = src.device
    runners = model.__dict__.setdefault("_graph_runners", {})
    key = (T, B, str(dev), float(model.dropout.p), int(_lib.load().rd_get_precision()))
    r = runners.get(key)
    if r is False:                                                     # capture failed before for this key: do not retry every call
        return None
    if r is not None and r.stale():
        r = None
    if r is None:
        runners.pop(key, None)
        # a runner keeps a whole step's activations: a loop that varies its batch size keeps the most recent few, not all of them
        live = [k for k, v in runners.items() if v is not False]
        while len(live) >= _max_runners():
            old = next((k for k in live if not runners[k].busy()), None)
            if old is None:                                            # every kept runner waits for a backward: this call goes the eager way
                return None
            live.remove(old)
            del runners[old]
        try:
            r = _Runner(model, T, B, dev)
        except (_lib.RaindropHipError, RuntimeError) as e:             # outside the captured step's envelope, or the capture itself failed
            import warnings
            warnings.warn("raindrop_amd.graph_module: no captured step for T=%d B=%d (%s: %s); these calls run operator by operator"
                          % (T, B, type(e).__name__, str(e).splitlines()[0] if str(e) else ""))
            runners[key] = False
            return None
        runners[key] = r
    if r.busy():                                                       # the previous captured call may still be backpropagated: leave its activations alone
        r.busy_hits = getattr(r, "busy_hits", 0) + 1
        if r.busy_hits == 8:                                           # e.g. a loop that appends un-detached logits to a list: say so once
            import warnings
            warnings.warn("raindrop_amd.graph_module: 8 training forwards in a row found the previous captured call still waiting for "
                          "its backward (its output is kept alive without loss.backward()): these calls run operator by operator, "
                          "~3x slower.  Detach what you keep (outputs.detach()), or set model.graph_step = False to silence this.")
        return None
    r.busy_hits = 0
    return _GraphStep.apply(r, src, static, times, lengths, *r.params)
