"""Oracle groundwork for SURVEY 8f rank 4 (host preprocessing): the restatement `oracle/preprocess.py`
against the fixture produced by the reference's own functions, and -- when the reference tree is present --
against those functions live on fresh data."""
import numpy as np
import pytest
import torch

from oracle import preprocess as O
from oracle import ref_loader
from tests.helpers import GOLDEN as GOLDEN_DIR


@pytest.fixture(scope="module")
def g():
    import os
    return np.load(os.path.join(GOLDEN_DIR, "preprocess.npz"))


def test_restatement_matches_reference_fixture(g):
    mf, stdf = O.get_stats(g["arr"].copy())
    ms, ss = O.get_stats_static(g["static"])
    assert np.array_equal(mf, g["mf"]) and np.array_equal(stdf, g["stdf"])
    assert np.array_equal(ms, g["ms"]) and np.array_equal(ss, g["ss"])          # the (0, 1) quirk
    P, Ps, Pt, y = O.tensorize_normalize(g["arr"].copy(), g["time"], g["static"].copy(), g["y"], mf, stdf, ms, ss)
    assert P.dtype == np.float32 and np.array_equal(P, g["P"])
    assert np.array_equal(Ps, g["Pstatic"]) and np.array_equal(Pt, g["Ptime"]) and np.array_equal(y, g["ytensor"])
    P2, none, Pt2, y2 = O.tensorize_normalize_other(g["arr"].copy(), g["y"], mf, stdf)
    assert none is None and np.array_equal(P2, g["P_other"]) and np.array_equal(Pt2, g["Ptime_other"])


def test_documented_quirks(g):
    F = g["arr"].shape[2]
    P = g["P"]
    assert np.array_equal(P[:, :, F:] > 0, g["arr"] > 0)                       # mask half == raw > 0
    assert np.all(P[:, :, :F][g["arr"] == 0] == 0)                             # missing -> 0 after normalisation
    assert np.all(g["Pstatic"] >= 0) and np.array_equal(g["Pstatic"] > 0, g["static"] > 0)   # negatives zeroed, no scaling
    assert np.allclose(g["Pstatic"][g["static"] > 0], g["static"][g["static"] > 0].astype(np.float32))
    assert np.array_equal(g["Ptime"], (g["time"].astype(np.float32) / np.float32(60.0)))


def test_setting3_removal_matches_the_script_loop():
    """`code/Raindrop.py:218-226`: per sample, `np.random.choice(F, k, replace=False)` value channels zeroed."""
    rng = np.random.default_rng(2)
    N, T, F = 7, 5, 10
    P = rng.standard_normal((N, T, 2 * F)).astype(np.float32)
    ratio = 0.3
    ref = torch.from_numpy(P.copy())
    np.random.seed(9)
    k = round(ratio * F)
    for i, patient in enumerate(ref):                                           # the script's loop, literally
        idx = np.random.choice(F, k, replace=False)
        patient[:, idx] = torch.zeros(ref.shape[1], k)
        ref[i] = patient
    np.random.seed(9)
    got = O.remove_features_per_sample(P.copy(), ratio)
    assert np.array_equal(got, ref.numpy())
    assert np.array_equal(got[:, :, F:], P[:, :, F:])                          # mask half untouched
    assert ((got[:, :, :F] == 0).all(axis=1).sum(axis=1) >= k).all()
    same = O.remove_features_set(P.copy(), np.arange(F)[::-1], ratio)
    assert np.all(same[:, :, [9, 8, 7]] == 0) and np.array_equal(same[:, :, :7], P[:, :, :7])


@pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")
def test_restatement_matches_reference_functions_live():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_preprocess_golden as mk
    u = mk.ref_utils()
    arr, time, static, y = mk.make_raw(17, 20, 11, 7, 9)
    recs = [{"arr": arr[i].copy(), "time": time[i].copy(), "extended_static": static[i].copy()} for i in range(len(arr))]
    mf, stdf = u.getStats(arr.copy())
    ms, ss = u.getStats_static(static.copy(), dataset="P12")
    P, Ps, Pt, yt = u.tensorize_normalize(recs, y.copy(), mf, stdf, ms, ss)
    omf, ostd = O.get_stats(arr.copy())
    oms, oss = O.get_stats_static(static)
    oP, oPs, oPt, oy = O.tensorize_normalize(arr.copy(), time, static.copy(), y, omf, ostd, oms, oss)
    assert np.array_equal(omf, mf) and np.array_equal(ostd, stdf) and np.array_equal(oms, ms) and np.array_equal(oss, ss)
    assert np.array_equal(oP, P.numpy()) and np.array_equal(oPs, Ps.numpy())
    assert np.array_equal(oPt, Pt.numpy()) and np.array_equal(oy, yt.numpy())
    P2, _, Pt2, _ = u.tensorize_normalize_other(arr.copy(), y.copy(), mf, stdf)
    oP2, _, oPt2, _ = O.tensorize_normalize_other(arr.copy(), y, omf, ostd)
    assert np.array_equal(oP2, P2.numpy()) and np.array_equal(oPt2, Pt2.numpy())
