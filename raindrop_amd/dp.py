"""Data-parallel training support: one process per GPU, samples sharded along the batch axis,
ONE flat live-gradient buffer all-reduced over RCCL (xGMI) per step.

The reference is single-process; this is the new multi-GPU layer
BASELINE.json asks for.  Every stage of `Raindrop_v2.forward` is per-sample (SURVEY.md section
8e), so the only exchange is the weight-gradient sum.  Design points for MI355X:
  * dead parameters (72-92 % of the state_dict) never get a gradient and are
    excluded, so the buffer is 2.0 MB at P19 instead of 7.8 MB;
  * after backward the 35 gradient tensors are packed into the flat buffer with one batched copy
    (autograd's per-parameter accumulate kernels are avoided by dropping old gradients), the buffer
    is all-reduced in a few large buckets, and the optimizer reads the reduced values in place
    (each `param.grad` is a view of the buffer; with `flatten_parameters()` the weights are too);
  * at <= 8 MB per bucket a ring all-reduce on the 7x153 GB/s full mesh is latency-, not
    bandwidth-bound, so few, large buckets win; the step after backward is only the pack + the
    collective + one Adam kernel, so there is little left to overlap the collective with.
"""
import torch
import torch.distributed as dist


class FlatGradAllReduce:
    """Flat gradient buffer + (for world > 1) its all-reduce.

    Per step:  zero() -> forward/backward -> finish() -> optimizer.
    `zero()` drops the previous gradients (p.grad = None), so autograd simply hands over each fresh
    gradient tensor (no 35 read-modify-write accumulate kernels); `finish()` packs them into the flat
    buffer with ONE batched copy, all-reduces the buffer over RCCL in a few large buckets, and
    re-points every `p.grad` at its slice of the flat buffer, which is what the optimizer reads."""

    def __init__(self, params, process_group=None, n_buckets=2, average=True, force_collective=False):
        """params: list of (name, Parameter) that will receive gradients, in FORWARD order (raindrop_amd.synth.live_parameter_names:
        R_u, emb, ob_propagation*, encoder layers 0.., mlp_static).  The order is a contract for the overlapped form below
        (tail_start / allreduce_range_async, used by TrainStep's two-graph step): the tail of the buffer from the last encoder
        layer's in_proj_weight on must hold exactly the gradients that are final before the rest of the backward pass runs.
        TrainStep checks it; model.named_parameters() order violates it (R_u, ob_propagation* come last there).
        force_collective (testing): issue the collectives even in a one-rank group -- a single MI355X can then exercise the RCCL
        calls themselves (backend load, AVG, async handles next to hipGraph replays), which a world of one otherwise skips."""
        self.group = process_group
        self.average = average
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if force_collective and dist.is_initialized():
            self._force = True                                            # (the true world size still sets the average)
        else:
            self._force = False
        self.params = [p for _, p in params]
        self.names = [n for n, _ in params]
        # every slice starts on a 256-byte boundary: the kernels use 16-byte loads on weights and
        # biases whenever the pointer allows it (the padding floats stay zero and are harmless to
        # the all-reduce and to Adam)
        align = 64
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += (p.numel() + align - 1) // align * align
        total = off
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.slices = []
        self.views = []
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.slices.append((o, o + n))
            self.views.append(self.flat[o:o + n].view_as(p))
        # buckets: contiguous ranges of the flat buffer, balanced by bytes; at <= 8 MB a ring
        # all-reduce on the xGMI mesh is latency-bound, so few and large
        n_buckets = max(1, min(n_buckets, len(self.params)))
        target = total / n_buckets
        bounds = [0]
        acc, b = 0, 0
        for i, p in enumerate(self.params):
            if acc >= target * (b + 1) and b < n_buckets - 1:
                bounds.append(self.slices[i][0])
                b += 1
            acc = self.slices[i][1]
        bounds.append(total)
        self.bounds = bounds
        self.n_buckets = len(bounds) - 1
        self.flat_param = None
        for p, v in zip(self.params, self.views):
            p.grad = v

    def flatten_parameters(self):
        """Re-home the live parameters themselves in ONE flat buffer (each `p.data` becomes a view) and
        return a single Parameter over it whose `.grad` is the flat gradient buffer: the optimizer
        then updates 0.5 M weights with one elementwise kernel instead of a 35-tensor multi-tensor
        launch.  Adam is elementwise, so this is numerically identical to per-tensor Adam."""
        total = self.flat.numel()
        pbuf = torch.zeros(total, dtype=torch.float32, device=self.flat.device)
        for p, (lo, hi) in zip(self.params, self.slices):
            pbuf[lo:hi].copy_(p.data.reshape(-1))
            p.data = pbuf[lo:hi].view_as(p)
        self.flat_param = torch.nn.Parameter(pbuf, requires_grad=True)
        self.flat_param.grad = self.flat
        return self.flat_param

    # -- per-step protocol -------------------------------------------------------------------
    def zero(self):
        """Forget last step's gradients (replaces optimizer.zero_grad(set_to_none=True))."""
        for p in self.params:
            p.grad = None

    def finish(self):
        """Pack the fresh gradients into the flat buffer (one batched copy), all-reduce it and leave
        every p.grad pointing at its slice.  Parameters without a gradient this step contribute 0."""
        grads = [(p.grad if p.grad is not None else torch.zeros_like(p)) for p in self.params]
        torch._foreach_copy_(self.views, grads)                  # one batched copy into the (padded) slices
        if self.world > 1:
            use_avg = self.average and dist.get_backend(self.group) == "nccl"
            op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
            handles = [dist.all_reduce(self.flat[self.bounds[b]:self.bounds[b + 1]], op=op, group=self.group,
                                       async_op=True) for b in range(self.n_buckets)]
            for h in handles:
                h.wait()
            if self.average and not use_avg:
                self.flat.div_(self.world)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def allreduce(self):
        """All-reduce the flat buffer in place (gradients were written into it directly, e.g. by
        raindrop_amd.step.TrainStep); no-op for a single process."""
        if self.world > 1 or self._force:
            use_avg = self.average and dist.get_backend(self.group) == "nccl"
            op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
            handles = [dist.all_reduce(self.flat[self.bounds[b]:self.bounds[b + 1]], op=op, group=self.group,
                                       async_op=True) for b in range(self.n_buckets)]
            for h in handles:
                h.wait()
            if self.average and not use_avg:
                self.flat.div_(self.world)

    # -- overlapped form: the tail of the buffer (parameters from `first_name` on, in forward order) is complete once the
    #    backward pass has left the last encoder layer; its collective runs beside the rest of the backward pass ---------------
    def tail_start(self, first_name):
        """Offset of parameter `first_name` in the flat buffer (slices start on 256-byte boundaries)."""
        return self.slices[self.names.index(first_name)][0]

    def allreduce_range_async(self, lo, hi):
        """Start the all-reduce of flat[lo:hi] on the collective's own stream (it waits for what the current stream has enqueued so
        far, and nothing later); returns a handle for allreduce_wait().  No-op (None) for a single process."""
        if (self.world <= 1 and not self._force) or hi <= lo:
            return None
        use_avg = self.average and dist.get_backend(self.group) == "nccl"
        op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
        return (dist.all_reduce(self.flat[lo:hi], op=op, group=self.group, async_op=True), lo, hi, use_avg)

    def allreduce_wait(self, handle):
        if handle is None:
            return
        h, lo, hi, use_avg = handle
        h.wait()
        if self.average and not use_avg:
            self.flat[lo:hi].div_(self.world)

    def nbytes(self):
        return self.flat.numel() * 4


def shard_batch(batch, rank, world):
    """Contiguous shard of a global batch along B: src[:, lo:hi], times[:, lo:hi], static[lo:hi] ..."""
    B = batch["src"].shape[1]
    assert B % world == 0, "global batch must divide evenly (equal shards keep CE-mean exact)"
    lo, hi = rank * (B // world), (rank + 1) * (B // world)
    out = dict(src=batch["src"][:, lo:hi].contiguous(), times=batch["times"][:, lo:hi].contiguous(),
               static=None if batch["static"] is None else batch["static"][lo:hi].contiguous(),
               lengths=batch["lengths"][lo:hi].contiguous(), y=batch["y"][lo:hi].contiguous())
    return out


def broadcast_parameters(module, src=0, group=None):
    """Identical replicas: rank `src`'s parameters and buffers overwrite everyone else's."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
