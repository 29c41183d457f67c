"""CPU: the synthetic dataset writer produces files the REFERENCE'S OWN loader and normaliser accept
(`code/utils_rd.py:23-146,221-257`), with the tensor layout the training loop then feeds to the
model (`code/Raindrop.py:232-238,310-317`).  Needs the reference tree; skipped on the GPU box."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import ref_loader
from raindrop_amd import compat_runner, synth

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")


def _utils_rd():
    path = os.path.join(ref_loader.reference_root(), "code", "utils_rd.py")
    spec = importlib.util.spec_from_file_location("ref_utils_rd", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("dataset", ["P19", "P12"])
def test_reference_loader_accepts_synthetic_dataset(tmp_path, dataset):
    u = _utils_rd()
    cfg = synth.make_config(dataset)
    n = 60
    base = compat_runner.write_dataset(str(tmp_path), dataset, n, seed=1)
    split = "/splits/" + compat_runner.DATASETS[dataset][3] % 1
    Ptrain, Pval, Ptest, ytrain, yval, ytest = u.get_data_split(base, split, split_type="random", reverse=False,
                                                                baseline=False, dataset=dataset)
    assert len(Ptrain) + len(Pval) + len(Ptest) == n
    T, F = Ptrain[0]["arr"].shape
    assert (T, F) == (cfg["max_len"], cfg["d_inp"])
    D = len(Ptrain[0]["extended_static"])
    X = np.stack([p["arr"] for p in Ptrain]); S = np.stack([p["extended_static"] for p in Ptrain])
    mf, stdf = compat_runner.get_stats_numpy2(X)      # utils_rd.getStats itself breaks on numpy >= 1.24 (:160)
    ms, ss = u.getStats_static(S, dataset=dataset)
    P, Pstatic, Ptime, y = u.tensorize_normalize(Ptrain, ytrain, mf, stdf, ms, ss)
    # Raindrop.py:232-238: permute to [T,N,2F] / [T,N]; then lengths = sum(Ptime > 0) (:317)
    P = P.permute(1, 0, 2); Ptime = Ptime.squeeze(2).permute(1, 0)
    assert tuple(P.shape) == (T, len(Ptrain), 2 * F) and tuple(Pstatic.shape) == (len(Ptrain), D)
    lengths = (Ptime > 0).sum(0)
    assert int(lengths.min()) >= 2 and int(lengths.max()) <= T
    # the observation-indicator half marks exactly the non-zero raw values (utils_rd.py:168,174)
    raw = np.stack([p["arr"] for p in Ptrain]).transpose(1, 0, 2)
    assert np.array_equal(P[:, :, F:].numpy() > 0, raw != 0)


def test_reference_loader_accepts_synthetic_pam_dataset(tmp_path):
    """PAM (BASELINE.json configs[0]): bare [T,F] records, no static features, 8 classes; the script's PAM branch
    (code/Raindrop.py:196-204) normalises with `tensorize_normalize_other` (code/utils_rd.py:243-257), which also invents the time
    axis linspace(0, T, T) / 60 -- first stamp 0, so `lengths` (Raindrop.py:317) is T - 1 for every sample."""
    u = _utils_rd()
    cfg = synth.make_config("PAM")
    n = 24
    base = compat_runner.write_dataset(str(tmp_path), "PAM", n, seed=2)
    split = "/splits/" + compat_runner.DATASETS["PAM"][3] % 1
    Ptrain, Pval, Ptest, ytrain, yval, ytest = u.get_data_split(base, split, split_type="random", reverse=False,
                                                                baseline=False, dataset="PAM")
    assert len(Ptrain) + len(Pval) + len(Ptest) == n
    T, F = Ptrain[0].shape
    assert (T, F) == (cfg["max_len"], cfg["d_inp"])
    raw = np.array(Ptrain, dtype=np.float64).transpose(1, 0, 2).copy()   # (mask_normalize works in place on the PAM records)
    mf, stdf = compat_runner.get_stats_numpy2(np.asarray(Ptrain, dtype=np.float64))
    P, Pstatic, Ptime, y = u.tensorize_normalize_other(Ptrain, ytrain, mf, stdf)
    assert Pstatic is None
    P = P.permute(1, 0, 2); Ptime = Ptime.squeeze(2).permute(1, 0)
    assert tuple(P.shape) == (T, len(Ptrain), 2 * F) and tuple(Ptime.shape) == (T, len(Ptrain))
    lengths = (Ptime > 0).sum(0)
    assert int(lengths.min()) == T - 1 and int(lengths.max()) == T - 1
    assert int(y.min()) >= 0 and int(y.max()) < cfg["n_classes"]
    assert np.array_equal(P[:, :, F:].numpy() > 0, raw != 0)


def test_workspace_layout(tmp_path):
    code = compat_runner.make_workspace(str(tmp_path), "P19", 40, ref_loader.reference_root())
    assert os.path.islink(os.path.join(code, "Raindrop.py")) and os.path.islink(os.path.join(code, "utils_rd.py"))
    shim = open(os.path.join(code, "models_rd.py")).read()
    assert "raindrop_amd.models_rd" in shim
    assert os.path.isdir(os.path.join(str(tmp_path), "models"))
    for k in range(1, 6):
        assert os.path.isfile(os.path.join(str(tmp_path), "P19data", "splits", "phy19_split%d_new.npy" % k))
