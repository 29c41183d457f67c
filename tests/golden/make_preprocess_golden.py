"""Generate tests/golden/preprocess.npz from the REFERENCE'S OWN preprocessing functions
(`/root/reference/code/utils_rd.py`, executed unmodified; only `getStats` gets the numpy-2-safe scalar max of
`raindrop_amd.compat_runner.get_stats_numpy2`, which is value-identical).  Run here, where the tree exists:

    python tests/golden/make_preprocess_golden.py
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import preprocess as O          # noqa: E402
from oracle import ref_loader               # noqa: E402
from raindrop_amd import compat_runner      # noqa: E402


def ref_utils():
    path = os.path.join(ref_loader.reference_root(), "code", "utils_rd.py")
    spec = importlib.util.spec_from_file_location("ref_utils_rd", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.getStats = compat_runner.get_stats_numpy2
    return mod


def make_raw(seed, N, T, F, D):
    rng = np.random.default_rng(seed)
    obs = rng.random((N, T, F)) < 0.3
    lens = rng.integers(2, T + 1, size=N)
    obs &= (np.arange(T)[None, :, None] < lens[:, None, None])
    arr = np.where(obs, np.abs(rng.standard_normal((N, T, F))) * 3 + 0.5, 0.0)
    time = np.zeros((N, T, 1))
    for i in range(N):
        time[i, :lens[i], 0] = np.cumsum(rng.uniform(1.0, 60.0, size=lens[i]))
    static = rng.standard_normal((N, D)) * 2
    y = (rng.random((N, 1)) < 0.3).astype(np.float64)
    return arr, time, static, y


def main():
    u = ref_utils()
    N, T, F, D = 12, 9, 5, 6
    arr, time, static, y = make_raw(0, N, T, F, D)
    recs = [{"arr": arr[i].copy(), "time": time[i].copy(), "extended_static": static[i].copy()} for i in range(N)]
    mf, stdf = u.getStats(arr.copy())
    ms, ss = u.getStats_static(static.copy(), dataset="P19")
    P, Ps, Pt, yt = u.tensorize_normalize(recs, y.copy(), mf, stdf, ms, ss)
    P2, _, Pt2, yt2 = u.tensorize_normalize_other(arr.copy(), y.copy(), mf, stdf)
    # the restatement must agree before anything is written
    omf, ostd = O.get_stats(arr.copy())
    oms, oss = O.get_stats_static(static)
    oP, oPs, oPt, oy = O.tensorize_normalize(arr.copy(), time, static.copy(), y, omf, ostd, oms, oss)
    assert np.array_equal(omf, mf) and np.array_equal(ostd, stdf) and np.array_equal(oms, ms) and np.array_equal(oss, ss)
    assert np.array_equal(oP, P.numpy()) and np.array_equal(oPs, Ps.numpy()) and np.array_equal(oPt, Pt.numpy())
    assert np.array_equal(oy, yt.numpy())
    out = os.path.join(ROOT, "tests", "golden", "preprocess.npz")
    np.savez_compressed(out, arr=arr, time=time, static=static, y=y, mf=mf, stdf=stdf, ms=ms, ss=ss,
                        P=P.numpy(), Pstatic=Ps.numpy(), Ptime=Pt.numpy(), ytensor=yt.numpy(),
                        P_other=P2.numpy(), Ptime_other=Pt2.numpy())
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
