"""CPU, end to end: the reference's UNMODIFIED training script (`code/Raindrop.py`, executed byte-for-byte by
`raindrop_amd.compat_runner.run`) on a synthetic P19 dataset written in the reference's on-disk format -- all five
splits x 20 epochs, ReduceLROnPlateau, validation, checkpoint save + load, test metrics.  No GPU exists in the build
container, so the `models_rd` the script imports is a shim over the REFERENCE'S OWN model (oracle O1, CPU patches of
oracle/ref_loader.py); everything else -- workspace, dataset files, compat patches, the script -- is exactly what the
HIP run uses (the HIP model itself is checked against the same reference model by the golden tests, and its training
loop on the GPU by tests/test_train_loop_gpu.py).  Needs the reference tree; skipped on the GPU box."""
import os

import numpy as np
import pytest

from oracle import ref_loader
from raindrop_amd import compat_runner

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")

ORACLE_SHIM = ("# test-only shim: the reference's own model (oracle O1) behind the module name the script imports\n"
               "from oracle import ref_loader as _rl\n"
               "_m = _rl.load().models_rd\n"
               "Raindrop_v2 = _m.Raindrop_v2\n"
               "PositionalEncodingTF = _m.PositionalEncodingTF\n"
               "__all__ = ['Raindrop_v2', 'PositionalEncodingTF']\n")


def test_unmodified_script_runs_end_to_end(tmp_path, capsys):
    import torch
    torch.set_num_threads(min(4, os.cpu_count() or 1))      # (all cores: 109 s alone, but 10 x that on a busy host -- OpenMP spin-waits; 4: ~130 s either way)
    root = str(tmp_path)
    with ref_loader._patched():                      # .cuda() -> identity etc.: the script is CUDA-only (Raindrop.py:253,310)
        g = compat_runner.run(root, "P19", 200, ref_loader.reference_root(), seed=3, model_shim=ORACLE_SHIM)
    out = capsys.readouterr().out
    assert "Dataset used:  P19" in out
    assert out.count("- - Run 1 - -") == 5                                   # five splits (Raindrop.py:152)
    assert out.count("Validation: Epoch") == 5 * 20                          # 20 epochs each, validated every epoch
    for k in range(1, 6):                                                    # checkpoints written and re-loaded (:374,381)
        assert os.path.isfile(os.path.join(root, "models", "raindrop_%d.pt" % k))
    acc = np.asarray(g["acc_arr"]); auc = np.asarray(g["auroc_arr"])
    assert acc.shape == (5, 1) and np.all(np.isfinite(acc)) and np.all((auc >= 0) & (auc <= 100))
