"""GPU: the device preprocessing (raindrop_amd/preprocess.py over rd_prep_*) against the fixture produced by the
reference's own functions (tests/golden/preprocess.npz) and against the numpy restatement (oracle/preprocess.py, itself
pinned to the reference) on larger inputs -- BIT-EXACT, statistics included: the kernels reproduce numpy's pairwise
summation tree, which only shows beyond 128 observed values per sensor (the fixture has ~30)."""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import preprocess as O
from raindrop_amd import preprocess as D
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu


def _np(t):
    return t.cpu().numpy()


def test_fixture_from_the_reference_bit_exact():
    g = np.load(os.path.join(GOLDEN, "preprocess.npz"))
    mf, stdf = D.getStats(g["arr"])
    assert np.array_equal(_np(mf), g["mf"]) and np.array_equal(_np(stdf), g["stdf"])
    ms, ss = D.getStats_static(g["static"], dataset="P19")
    assert np.array_equal(ms, g["ms"]) and np.array_equal(ss, g["ss"])
    N = len(g["arr"])
    recs = [{"arr": g["arr"][i], "time": g["time"][i], "extended_static": g["static"][i]} for i in range(N)]
    P, Ps, Pt, y = D.tensorize_normalize(recs, g["y"], mf, stdf, ms, ss)
    assert P.dtype == torch.float32 and np.array_equal(_np(P), g["P"])
    assert np.array_equal(_np(Ps), g["Pstatic"]) and np.array_equal(_np(Pt), g["Ptime"]) and np.array_equal(_np(y), g["ytensor"])
    # the training loop's layout (code/Raindrop.py:232-238), written directly
    P2, _, Pt2, _ = D.tensorize_normalize((g["arr"], g["time"], g["static"]), g["y"], mf, stdf, ms, ss, time_major=True)
    assert np.array_equal(_np(P2), g["P"].transpose(1, 0, 2)) and np.array_equal(_np(Pt2), g["Ptime"][:, :, 0].T)
    Po, none, Pto, yo = D.tensorize_normalize_other(g["arr"], g["y"], mf, stdf)
    assert none is None and np.array_equal(_np(Po), g["P_other"]) and np.array_equal(_np(Pto), g["Ptime_other"])


@pytest.mark.parametrize("N,T,F,seed", [(700, 60, 7, 1), (333, 215, 5, 2), (4000, 60, 3, 3)])
def test_statistics_reproduce_numpy_pairwise_sums(N, T, F, seed):
    """Thousands to 10^5 observed values per sensor: numpy's add-reduction splits them recursively into <= 128-element
    blocks; mean / std must come out bit-identical, and so must everything normalised with them."""
    rng = np.random.default_rng(seed)
    dens = np.linspace(0.9, 0.02, F)
    arr = np.where(rng.random((N, T, F)) < dens[None, None, :], np.abs(rng.standard_normal((N, T, F))) * 37.5 + 0.01, 0.0)
    arr[:, :, F - 1] = 0.0
    arr[0, :5, F - 1] = [3.0, 1.5, 2.25, 8.0, 0.125]                    # a sensor with < 8 observed values (plain loop branch)
    mf_ref, std_ref = O.get_stats(arr.copy())
    mf, stdf = D.getStats(arr)
    assert np.array_equal(_np(mf), mf_ref), np.abs(_np(mf) - mf_ref).max()
    assert np.array_equal(_np(stdf), std_ref), np.abs(_np(stdf) - std_ref).max()
    want = O.mask_normalize(arr.astype(np.float64), mf_ref, std_ref).astype(np.float32)
    assert np.array_equal(_np(D.mask_normalize(arr, mf, stdf)), want)
    assert np.array_equal(_np(D.mask_normalize(arr, mf, stdf, time_major=True)), want.transpose(1, 0, 2))


def test_empty_sensor_gives_nan_like_numpy():
    arr = np.zeros((4, 6, 2)); arr[:, :, 0] = 2.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mf_ref, std_ref = O.get_stats(arr.copy())
    mf, stdf = D.getStats(arr)
    assert np.array_equal(_np(mf), mf_ref, equal_nan=True) and np.array_equal(_np(stdf), std_ref, equal_nan=True)
    assert np.isnan(_np(mf)[1, 0]) and _np(stdf)[0, 0] == 1e-7           # constant sensor: std 0 -> floor


def test_static_and_time():
    rng = np.random.default_rng(4)
    S = rng.standard_normal((50, 9)) * 3
    ms, ss = D.getStats_static(S, "P12")
    assert np.array_equal(_np(D.mask_normalize_static(S, ms, ss)), O.mask_normalize_static(S.copy(), ms, ss).astype(np.float32))


def test_feature_removal_settings_2_and_3():
    rng = np.random.default_rng(6)
    N, T, F = 9, 7, 10
    P = rng.standard_normal((N, T, 2 * F)).astype(np.float32)
    ratio = 0.3
    k = round(ratio * F)
    np.random.seed(5)
    want = O.remove_features_per_sample(P.copy(), ratio)                  # the script's per-patient draws, in order
    np.random.seed(5)
    idx = np.stack([np.random.choice(F, k, replace=False) for _ in range(N)])
    got = D.remove_features(torch.from_numpy(P.copy()).cuda(), idx)
    assert np.array_equal(_np(got), want)
    got_tm = D.remove_features(torch.from_numpy(P.transpose(1, 0, 2).copy()).cuda(), idx, time_major=True)
    assert np.array_equal(_np(got_tm), want.transpose(1, 0, 2))
    ranked = np.arange(F)[::-1]
    want2 = O.remove_features_set(P.copy(), ranked, ratio)
    got2 = D.remove_features(torch.from_numpy(P.copy()).cuda(), ranked[:k].copy())
    assert np.array_equal(_np(got2), want2)


def test_no_cpu_path():
    from raindrop_amd import _lib
    with pytest.raises(_lib.RaindropHipError):
        D.getStats(np.ones((2, 3, 4)), device="cpu")
