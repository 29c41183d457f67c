"""Batch feed from a device-resident dataset.

The reference slices every training batch on the host and copies it to the GPU
(`code/Raindrop.py:310-315`: `Ptrain_tensor[:, idx, :].cuda()` ... four fancy-index + H2D transfers per
step, 4.4 MB at P19/B=256) and pushes the WHOLE validation split through one forward
(`code/utils_rd.py:310-320`).  With the hot path at ~1 ms per step that host work dominates, so the
dataset is kept in HBM and a batch is one `rd_batch_gather` launch (bit-exact copies + `lengths`):

    ds = DeviceDataset(Ptrain_tensor, Ptrain_time_tensor, Ptrain_static_tensor, ytrain_tensor)
    for idx in epoch_index_plan(ytrain, batch_size=128, strategy=2):        # Raindrop.py:262-307
        P, Pstatic, Ptime, y, lengths = ds.batch(idx)                        # Raindrop.py:310-317
        outputs, _, _ = model.forward(P, Pstatic, Ptime, lengths)
    logits = evaluate_chunked(model, val_ds)                                 # utils_rd.py:310-320

The index plan is the reference's own host logic (same numpy calls in the same order, so the same
`np.random.seed` gives the same batches); only the data movement changed.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


class DeviceDataset:
    """P [T,N,2F] f32, Ptime [T,N] f32, Pstatic [N,d_static] f32 or None, y [N] int64 or None -- moved to
    `device` once.  `batch(idx)` returns freshly allocated (P, Pstatic, Ptime, y, lengths) for the index
    vector `idx` (numpy / list / tensor), or fills the buffers of a previous call when `out=` is given."""

    def __init__(self, P, Ptime, Pstatic=None, y=None, device="cuda"):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.RaindropHipError("DeviceDataset needs a ROCm device (there is no CPU fallback)")
        self.dev = dev
        self.P = torch.as_tensor(P, dtype=torch.float32).to(dev).contiguous()
        self.Ptime = torch.as_tensor(Ptime, dtype=torch.float32).to(dev).contiguous()
        self.Pstatic = None if Pstatic is None else torch.as_tensor(Pstatic, dtype=torch.float32).to(dev).contiguous()
        self.y = None if y is None else torch.as_tensor(y).reshape(-1).to(torch.int64).to(dev).contiguous()
        self.T, self.N, self.W = self.P.shape
        if tuple(self.Ptime.shape) != (self.T, self.N):
            raise ValueError("Ptime must be [T,N] = [%d,%d], got %s" % (self.T, self.N, tuple(self.Ptime.shape)))
        self.ds = 0 if self.Pstatic is None else int(self.Pstatic.shape[1])
        if self.Pstatic is not None and self.Pstatic.shape[0] != self.N:
            raise ValueError("Pstatic must have N=%d rows" % self.N)
        if self.y is not None and self.y.numel() != self.N:
            raise ValueError("y must have N=%d entries" % self.N)
        self._bad = torch.zeros(1, dtype=torch.int32, device=dev)

    def __len__(self):
        return self.N

    def alloc(self, B):
        f32 = dict(dtype=torch.float32, device=self.dev)
        return {"P": torch.empty((self.T, B, self.W), **f32), "Ptime": torch.empty((self.T, B), **f32),
                "Pstatic": None if self.Pstatic is None else torch.empty((B, self.ds), **f32),
                "y": None if self.y is None else torch.empty((B,), dtype=torch.int64, device=self.dev),
                "lengths": torch.empty((B,), dtype=torch.int64, device=self.dev)}

    def batch(self, idx, out=None, check=True):
        idx_t = torch.as_tensor(np.asarray(idx) if not torch.is_tensor(idx) else idx).reshape(-1).to(torch.int64)
        if check and idx_t.device.type == "cpu" and idx_t.numel():      # host indices: validate for free
            lo, hi = int(idx_t.min()), int(idx_t.max())
            if lo < 0 or hi >= self.N:
                raise IndexError("batch index out of range [0,%d): min %d max %d" % (self.N, lo, hi))
        idx_t = idx_t.to(self.dev)
        B = idx_t.numel()
        o = out if out is not None else self.alloc(B)
        if o["P"].shape[1] != B:
            raise ValueError("out buffers hold %d samples, idx has %d" % (o["P"].shape[1], B))
        _lib.call("rd_batch_gather", self.T, B, self.W, self.ds, self.N, _p(self.P), _p(self.Ptime), _p(self.Pstatic),
                  _p(self.y), _p(idx_t), _p(o["P"]), _p(o["Ptime"]), _p(o["Pstatic"]), _p(o["y"]), _p(o["lengths"]),
                  _p(self._bad), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return o["P"], o["Pstatic"], o["Ptime"], o["y"], o["lengths"]

    def bad_indices(self):
        """Number of out-of-range indices clamped by batch() since the last call of this method (device-side
        index vectors are not checked on the host); reading it synchronises."""
        n = int(self._bad.item())
        self._bad.zero_()
        return n


class EpochPlanner:
    """The reference's batch plan over a whole RUN (`code/Raindrop.py:261-307`): `idx_0` and the 3x-expanded `idx_1` are
    created once per run and shuffled IN PLACE at the start of every epoch, so epoch k's permutation is applied on top of
    epoch k-1's.  `next_epoch()` returns the batches of the next epoch; with the same `np.random.seed` the sequence of
    batches over ALL epochs equals the script's (`epoch_index_plan` rebuilds the arrays and therefore only reproduces the
    FIRST epoch of a run)."""

    def __init__(self, ytrain, batch_size=128, strategy=2, n_total=None):
        y = np.asarray(ytrain).reshape(-1)
        self.idx_0 = np.where(y == 0)[0]
        self.idx_1 = np.where(y == 1)[0]
        self.expanded = np.concatenate([self.idx_1, self.idx_1, self.idx_1], axis=0)
        self.batch_size, self.strategy = int(batch_size), int(strategy)
        self.n = len(y) if n_total is None else int(n_total)
        half = int(batch_size / 2)
        self.n_batches = {1: 10, 3: 30}.get(self.strategy) if self.strategy != 2 else \
            int(np.min([len(self.idx_0) // half, len(self.expanded) // half]))
        if self.strategy not in (1, 2, 3):
            raise ValueError("strategy must be 1, 2 or 3")

    def next_epoch(self):
        half = int(self.batch_size / 2)
        if self.strategy == 2:
            np.random.shuffle(self.expanded)                     # in place, cumulative over epochs (Raindrop.py:293-296)
            np.random.shuffle(self.idx_0)
            return [np.concatenate([self.idx_0[n * half:(n + 1) * half], self.expanded[n * half:(n + 1) * half]], axis=0)
                    for n in range(self.n_batches)]
        if self.strategy == 3:
            return [np.random.choice(list(range(self.n)), size=self.batch_size, replace=False) for _ in range(30)]
        return [np.concatenate([np.random.choice(self.idx_0, size=half, replace=False),
                                np.random.choice(self.idx_1, size=half, replace=False)], axis=0) for _ in range(10)]


def epoch_index_plan(ytrain, batch_size=128, strategy=2, n_total=None):
    """The batches of the FIRST epoch of a run as the reference draws them (`code/Raindrop.py:262-307`), using the global
    numpy RNG exactly like the script (call `np.random.seed` first for reproducibility).  Later epochs of the script
    shuffle the arrays of the previous epoch in place: use `EpochPlanner` for a multi-epoch run.
      strategy 2 (P12/P19): minority class repeated 3x, both classes shuffled (positives first, then
        negatives -- the order of the two `np.random.shuffle` calls matters), batch n = batch_size/2
        negatives ++ batch_size/2 positives, `min(n0, 3*n1) // (batch_size/2)` batches;
      strategy 3 (PAM): 30 batches of `batch_size` distinct samples (`np.random.choice(..., replace=False)`);
      strategy 1: 10 balanced batches drawn without replacement per class (`utils_rd.random_sample`)."""
    y = np.asarray(ytrain).reshape(-1)
    idx_0 = np.where(y == 0)[0]
    idx_1 = np.where(y == 1)[0]
    half = int(batch_size / 2)
    if strategy == 2:
        expanded = np.concatenate([idx_1, idx_1, idx_1], axis=0)
        n_batches = int(np.min([len(idx_0) // half, len(expanded) // half]))
        np.random.shuffle(expanded)
        np.random.shuffle(idx_0)
        return [np.concatenate([idx_0[n * half:(n + 1) * half], expanded[n * half:(n + 1) * half]], axis=0)
                for n in range(n_batches)]
    if strategy == 3:
        n = len(y) if n_total is None else int(n_total)
        return [np.random.choice(list(range(n)), size=int(batch_size), replace=False) for _ in range(30)]
    if strategy == 1:
        return [np.concatenate([np.random.choice(idx_0, size=half, replace=False),
                                np.random.choice(idx_1, size=half, replace=False)], axis=0) for _ in range(10)]
    raise ValueError("strategy must be 1, 2 or 3")


@torch.no_grad()
def evaluate_chunked(model, ds, chunk=2048):
    """`utils_rd.evaluate_standard` (`code/utils_rd.py:310-320`: the whole split through one forward) as
    contiguous chunks: every stage of the model is per-sample, so the concatenated logits are identical
    (tests/test_gpu_parity.py::test_batch_invariance_at_validation_scale) while the activations stay bounded."""
    was_training = model.training
    model.eval()
    outs = []
    try:
        for lo in range(0, ds.N, chunk):
            hi = min(ds.N, lo + chunk)
            P = ds.P[:, lo:hi].contiguous()
            Ptime = ds.Ptime[:, lo:hi].contiguous()
            Pstatic = None if ds.Pstatic is None else ds.Pstatic[lo:hi].contiguous()
            lengths = torch.sum(Ptime > 0, dim=0)
            out, _, _ = model.forward(P, Pstatic, Ptime, lengths)
            outs.append(out)
    finally:
        model.train(was_training)
    return torch.cat(outs, 0) if outs else torch.empty((0, 0), device=ds.dev)


@torch.no_grad()
def evaluate_sharded(model, ds, chunk=2048, group=None):
    """`utils_rd.evaluate_standard` (`code/utils_rd.py:310-320`) over a process group: rank r runs the
    contiguous shard [r*ceil(N/W), ...) of the split through `evaluate_chunked`'s per-chunk forward and the logits are
    all-gathered (RCCL over xGMI with the nccl backend) into the full [N, C] tensor on every rank -- identical to the
    single-process result because every stage of the model is per-sample.  Without an initialised process group this
    is `evaluate_chunked`."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return evaluate_chunked(model, ds, chunk)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (ds.N + world - 1) // world
    lo, hi = min(ds.N, rank * per), min(ds.N, (rank + 1) * per)
    was_training = model.training
    model.eval()
    outs = []
    try:
        for a in range(lo, hi, chunk):
            b = min(hi, a + chunk)
            P = ds.P[:, a:b].contiguous()
            Ptime = ds.Ptime[:, a:b].contiguous()
            Pstatic = None if ds.Pstatic is None else ds.Pstatic[a:b].contiguous()
            out, _, _ = model.forward(P, Pstatic, Ptime, torch.sum(Ptime > 0, dim=0))
            outs.append(out)
    finally:
        model.train(was_training)
    C = int(model.n_classes)
    mine = torch.zeros((per, C), dtype=torch.float32, device=ds.dev)          # equal-sized shards for the collective
    if outs:
        mine[: hi - lo] = torch.cat(outs, 0)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.cat(parts, 0)[: ds.N]
