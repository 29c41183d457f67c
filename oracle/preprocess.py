"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's host preprocessing.

Only `tests/` may import this module.  The device version is `raindrop_amd/preprocess.py` over `rd_prep_*`
(raindrop_amd/csrc/rd_preprocess.hip); `tests/test_preprocess_gpu.py` compares it with this file bit for bit.

Follows `code/utils_rd.py:149-257` (statistics, masking normalisation, tensorisation) and the Setting-2/3
feature removal of `code/Raindrop.py:215-231`, vectorised, with every quirk of the originals kept:
  * `getStats` uses only values > 0, the POPULATION standard deviation, and a 1e-7 floor (`:149-161`);
    the original's `np.max([stdf[f], eps])` is a ragged list on numpy >= 1.24 and raises -- same values here;
  * `mask_normalize` divides by `stdf + 1e-18` and multiplies by the mask `P > 0`, then appends the mask as
    F extra channels (`:164-175`);
  * `getStats_static` compares the LIST `bool_categorical` with 0 (`:199`), which is never true: it always
    returns mean 0 / std 1, i.e. static features are NOT normalised -- reproduced, not "fixed";
  * `mask_normalize_static` zeroes every entry <= 0 AFTER the (identity) normalisation (`:206-219`);
  * times are minutes / 60 (`:235`); PAM-style inputs get `linspace(0, T, T) / 60` (`:248-253`, float32 linspace);
  * results are float64 numpy arrays that the reference then casts with `torch.Tensor(...)` (float32).
Parity status: pinned by execution against the reference's own functions
(`tests/test_preprocess_oracle.py`, live when `/root/reference` is present) and by the committed fixture
`tests/golden/preprocess.npz` produced from them (`tests/golden/make_preprocess_golden.py`).
"""
import numpy as np


def get_stats(P):
    """P [N,T,F] raw (0 = missing) -> (mf [F,1], stdf [F,1])."""
    N, T, F = P.shape
    Pf = P.transpose(2, 0, 1).reshape(F, -1)
    mf, stdf = np.zeros((F, 1)), np.ones((F, 1))
    for f in range(F):
        v = Pf[f][Pf[f] > 0]
        mf[f] = np.mean(v)
        stdf[f] = max(float(np.std(v)), 1e-7)
    return mf, stdf


def mask_normalize(P, mf, stdf):
    """[N,T,F] -> [N,T,2F]: normalised values (0 where missing) ++ observation mask."""
    M = (P > 0).astype(np.int64)
    Pn = (P - mf.reshape(1, 1, -1)) / (stdf.reshape(1, 1, -1) + 1e-18) * M
    return np.concatenate([Pn, M], axis=2)


def get_stats_static(S):
    """Always (0, 1): see the module docstring (`bool_categorical == 0` compares a list with an int)."""
    n = S.shape[1]
    return np.zeros((n, 1)), np.ones((n, 1))


def mask_normalize_static(S, ms, ss):
    out = (S - ms.reshape(1, -1)) / (ss.reshape(1, -1) + 1e-18)
    out = np.where(out <= 0, 0.0, out)
    return out


def tensorize_normalize(arr, time_minutes, static, y, mf, stdf, ms, ss):
    """arr [N,T,F], time [N,T,1] minutes, static [N,D], y [N,1] -> (P [N,T,2F] f32, Pstatic [N,D] f32,
    Ptime [N,T,1] f32 hours, y [N] int64) -- the tuple of `utils_rd.tensorize_normalize` as numpy arrays."""
    P = mask_normalize(arr.astype(np.float64), mf, stdf).astype(np.float32)
    Pt = (time_minutes.astype(np.float32) / np.float32(60.0)).astype(np.float32)
    Ps = mask_normalize_static(static.astype(np.float64), ms, ss).astype(np.float32)
    return P, Ps, Pt, y[:, 0].astype(np.float32).astype(np.int64)


def tensorize_normalize_other(arr, y, mf, stdf):
    """PAM-style records (bare [T,F] arrays): time = float32 linspace(0,T,T) / 60, no static features."""
    import torch
    N, T, F = arr.shape
    tim = torch.linspace(0, T, T).reshape(-1, 1).numpy().astype(np.float64)        # float32 values, as stored
    Pt = (np.broadcast_to(tim, (N, T, 1)).astype(np.float32) / np.float32(60.0)).astype(np.float32)
    P = mask_normalize(arr.astype(np.float64), mf, stdf).astype(np.float32)
    return P, None, Pt, y[:, 0].astype(np.float32).astype(np.int64)


def remove_features_per_sample(P, missing_ratio):
    """Setting 3 ('sample' level, `code/Raindrop.py:215-226`): for every sample, zero
    `round(missing_ratio * F)` value channels chosen by `np.random.choice(F, k, replace=False)` (global numpy
    RNG, one draw per sample, in order).  P [N,T,2F] is modified in place like the original; the mask half is
    left untouched (the reference zeroes only indices < F)."""
    F = P.shape[2] // 2
    k = round(missing_ratio * F)
    for i in range(P.shape[0]):
        idx = np.random.choice(F, k, replace=False)
        P[i][:, idx] = 0
    return P


def remove_features_set(P, ranked_indices, missing_ratio):
    """Setting 2 ('set' level, `:227-231`): zero the same `round(missing_ratio * F)` top-ranked value channels
    for every sample."""
    F = P.shape[2] // 2
    k = round(missing_ratio * F)
    idx = np.asarray(ranked_indices)[:k].astype(int)
    P[:, :, idx] = 0
    return P
