"""Host preprocessing of the reference on the device: the `utils_rd` functions of
`code/utils_rd.py:149-257` and the Setting-2/3 feature removal of `code/Raindrop.py:215-231`, same names, same
argument meaning, results BIT-IDENTICAL to the reference's numpy/torch (tests/test_preprocess_gpu.py).

    mf, stdf = getStats(Ptrain_array)                                   # [F,1] float64, on the device
    ms, ss = getStats_static(static_array, dataset="P19")               # the reference's (0, 1) quirk, kept
    P, Pstatic, Ptime, y = tensorize_normalize(records, y, mf, stdf, ms, ss, time_major=True)
    remove_features(Pval, idx, time_major=True)                         # Setting 2 / 3

Inputs may be numpy arrays or tensors (moved to the device as float64: the reference computes in float64 numpy and
casts to float32 at the very end).  `time_major=True` writes P as [T,N,2F] and Ptime as [T,N] directly -- the layout
the training loop permutes to (`code/Raindrop.py:232-238`); the default reproduces `utils_rd`'s own [N,T,2F] / [N,T,1].
There is no CPU path: without a ROCm device these functions raise.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev64(a, device):
    t = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a)
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.RaindropHipError("raindrop_amd.preprocess needs a ROCm device (there is no CPU fallback)")
    return t.to(device=dev, dtype=torch.float64).contiguous()


def getStats(P_tensor, device="cuda"):
    """`utils_rd.getStats`: P [N,T,F] raw values (0 = missing) -> (mf [F,1], stdf [F,1]) float64 device tensors."""
    P = _dev64(P_tensor, device)
    N, T, F = P.shape
    mf = torch.empty((F, 1), dtype=torch.float64, device=P.device)
    stdf = torch.empty((F, 1), dtype=torch.float64, device=P.device)
    lib = _lib.load()
    ws = torch.empty(max(int(lib.rd_prep_stats_workspace_bytes(N * T, F)), 256), dtype=torch.uint8, device=P.device)
    _lib.call("rd_prep_stats", N * T, F, _p(P), _p(mf), _p(stdf), _p(ws), ws.numel(), _st())
    return mf, stdf


def getStats_static(P_tensor, dataset="P12"):
    """`utils_rd.getStats_static` (:189-203): compares the LIST `bool_categorical` with 0, which is never true, so it
    always returns mean 0 / std 1 -- static features are NOT normalised by the reference.  Reproduced, not fixed."""
    S = np.asarray(P_tensor.cpu() if torch.is_tensor(P_tensor) else P_tensor).shape[1]
    return np.zeros((S, 1)), np.ones((S, 1))


def mask_normalize(P_tensor, mf, stdf, device="cuda", time_major=False):
    """`utils_rd.mask_normalize` + the float32 cast of its caller: [N,T,F] -> float32 [N,T,2F] (or [T,N,2F])."""
    P = _dev64(P_tensor, device)
    N, T, F = P.shape
    mfd, sdd = _dev64(mf, P.device).reshape(-1), _dev64(stdf, P.device).reshape(-1)
    out = torch.empty((T, N, 2 * F) if time_major else (N, T, 2 * F), dtype=torch.float32, device=P.device)
    _lib.call("rd_prep_mask_normalize", N, T, F, _p(P), _p(mfd), _p(sdd), _p(out), 1 if time_major else 0, _st())
    return out


def mask_normalize_static(P_tensor, ms, ss, device="cuda"):
    S = _dev64(P_tensor, device)
    N, D = S.shape
    out = torch.empty((N, D), dtype=torch.float32, device=S.device)
    msd, ssd = _dev64(ms, S.device).reshape(-1), _dev64(ss, S.device).reshape(-1)     # named: alive until the launch is enqueued
    _lib.call("rd_prep_static", N, D, _p(S), _p(msd), _p(ssd), _p(out), _st())
    return out


def _times(minutes, device, time_major):
    tm = _dev64(minutes, device)
    N, T = tm.shape[0], tm.shape[1]
    out = torch.empty((T, N) if time_major else (N, T, 1), dtype=torch.float32, device=tm.device)
    _lib.call("rd_prep_time", N, T, _p(tm), _p(out), 1 if time_major else 0, _st())
    return out


def tensorize_normalize(P, y, mf, stdf, ms, ss, device="cuda", time_major=False):
    """`utils_rd.tensorize_normalize` (:221-241).  P: list of records {'arr' [T,F], 'time' [T,1] minutes,
    'extended_static' [D]} (the reference's on-disk format) or a tuple (arr [N,T,F], time [N,T,1], static [N,D]).
    Returns (P_tensor, P_static_tensor, P_time, y_tensor) like the reference, on the device."""
    if isinstance(P, tuple):
        arr, tim, sta = P
    else:
        arr = np.stack([p["arr"] for p in P]); tim = np.stack([p["time"] for p in P])
        sta = np.stack([p["extended_static"] for p in P])
    Pt = mask_normalize(arr, mf, stdf, device, time_major)
    Ptime = _times(np.asarray(tim).reshape(len(arr), -1), device, time_major)
    Ps = mask_normalize_static(sta, ms, ss, device)
    yv = np.asarray(y)[:, 0]
    yt = torch.as_tensor(yv.astype(np.float32)).to(torch.int64).to(Pt.device)      # torch.Tensor(y[:,0]).type(LongTensor)
    return Pt, Ps, Ptime, yt


def tensorize_normalize_other(P, y, mf, stdf, device="cuda", time_major=False):
    """`utils_rd.tensorize_normalize_other` (:243-257, PAM-style bare [T,F] records): time = float32
    `linspace(0, T, T)` (stored through a float64 array, as the reference does) / 60, no static features."""
    arr = np.asarray(P)
    N, T, F = arr.shape
    tim = torch.linspace(0, T, T).reshape(1, -1).to(torch.float64).expand(N, T)
    Pt = mask_normalize(arr, mf, stdf, device, time_major)
    Ptime = _times(tim, device, time_major)
    yv = np.asarray(y)[:, 0]
    return Pt, None, Ptime, torch.as_tensor(yv.astype(np.float32)).to(torch.int64).to(Pt.device)


def remove_features(P_tensor, idx, time_major=False):
    """Setting 2 / 3 (`code/Raindrop.py:215-231`), in place on a float32 device tensor [N,T,2F] (or [T,N,2F]):
    idx [N,k] -> per-sample channels (the reference draws them with one `np.random.choice(F, k, replace=False)` per
    patient, in order: draw them on the host the same way and pass them here); idx [k] -> the same set for everybody."""
    if not (torch.is_tensor(P_tensor) and P_tensor.is_cuda and P_tensor.dtype == torch.float32 and P_tensor.is_contiguous()):
        raise _lib.RaindropHipError("remove_features needs a contiguous float32 device tensor")
    if time_major:
        T, N, W = P_tensor.shape
    else:
        N, T, W = P_tensor.shape
    ix = torch.as_tensor(np.asarray(idx)).to(torch.int32).to(P_tensor.device).contiguous()
    per_sample = 1 if ix.dim() == 2 else 0
    k = ix.shape[-1]
    if per_sample and ix.shape[0] != N:
        raise ValueError("idx must be [N,k] or [k]")
    _lib.call("rd_prep_remove_features", N, T, W // 2, _p(P_tensor), _p(ix), k, per_sample, 1 if time_major else 0, _st())
    return P_tensor
