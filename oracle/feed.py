"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's batch feed.

Only `tests/` may import this module; the product (`raindrop_amd/feed.py`) calls `rd_batch_gather` through
the C-ABI and raises without a ROCm device.

`gather_batch`  -- `code/Raindrop.py:310-317`: numpy/torch fancy indexing of the four dataset tensors and
                   `lengths = torch.sum(Ptime > 0, dim=0)`.
`epoch_batches` -- `code/Raindrop.py:262-307`, written as the literal per-batch loop of the script.
Parity status: pinned by execution -- `tests/test_feed.py::test_index_plan_matches_reference_script_lines`
runs the reference's own lines (extracted at test time from `/root/reference/code/Raindrop.py` when the tree is
present) against this restatement; the gather has no arithmetic (index expressions on the same arrays).
"""
import numpy as np


def gather_batch(P, Ptime, Pstatic, y, idx):
    idx = np.asarray(idx)
    Pb = P[:, idx, :]                                   # Raindrop.py:311
    Tb = Ptime[:, idx]
    Sb = None if Pstatic is None else Pstatic[idx]
    yb = None if y is None else y[idx]
    lengths = (Tb > 0).sum(axis=0).astype(np.int64)     # Raindrop.py:317
    return Pb, Sb, Tb, yb, lengths


def epoch_batches(ytrain, batch_size, strategy, n_total=None):
    ytrain = np.asarray(ytrain).reshape(-1)
    idx_0 = np.where(ytrain == 0)[0]                    # Raindrop.py:261-262
    idx_1 = np.where(ytrain == 1)[0]
    n0 = len(idx_0)
    expanded_idx_1 = np.concatenate([idx_1, idx_1, idx_1], axis=0)      # :270
    expanded_n1 = len(expanded_idx_1)
    if strategy == 1:
        n_batches = 10
    elif strategy == 2:
        K0 = n0 // int(batch_size / 2)
        K1 = expanded_n1 // int(batch_size / 2)
        n_batches = np.min([K0, K1])
    else:
        n_batches = 30
    if strategy == 2:                                   # :292-296
        np.random.shuffle(expanded_idx_1)
        I1 = expanded_idx_1
        np.random.shuffle(idx_0)
        I0 = idx_0
    out = []
    for n in range(n_batches):                          # :298-308
        if strategy == 1:
            idx0_batch = np.random.choice(idx_0, size=int(batch_size / 2), replace=False)
            idx1_batch = np.random.choice(idx_1, size=int(batch_size / 2), replace=False)
            idx = np.concatenate([idx0_batch, idx1_batch], axis=0)
        elif strategy == 2:
            idx0_batch = I0[n * int(batch_size / 2):(n + 1) * int(batch_size / 2)]
            idx1_batch = I1[n * int(batch_size / 2):(n + 1) * int(batch_size / 2)]
            idx = np.concatenate([idx0_batch, idx1_batch], axis=0)
        else:
            N = len(ytrain) if n_total is None else n_total
            idx = np.random.choice(list(range(N)), size=int(batch_size), replace=False)
        out.append(idx)
    return out
