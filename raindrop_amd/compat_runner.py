"""Run the reference's UNMODIFIED training script (`code/Raindrop.py`) on this model with synthetic data.

    python -m raindrop_amd.compat_runner --dataset P19 --samples 2000 [--reference /root/reference]

The reference ships neither its datasets (`.MISSING_LARGE_BLOBS`) nor a torch-2.x-compatible
script, so "runs unchanged" means: the script file is executed byte-for-byte
via `runpy`, inside a scratch workspace that provides
  * `<ws>/<X>data/processed_data/*.npy`, `<ws>/<X>data/splits/*.npy` -- synthetic records in the
    reference's on-disk format (`P12data/process_scripts/IrregularSampling.py:69-92`: dicts with
    `arr [T,F]`, `time [T,1]` in minutes, `extended_static`, `length`; loader `code/utils_rd.py:23-146`),
  * `<ws>/code/{Raindrop.py,utils_rd.py}` as symlinks into the reference tree (never copied),
  * `<ws>/code/models_rd.py` -- a two-line shim re-exporting `raindrop_amd.models_rd`,
  * `<ws>/models/` for the checkpoint the script writes (`code/Raindrop.py:74,374`),
and with process-level compat patches that do not touch the script: `ReduceLROnPlateau` drops the
removed `verbose` kwarg (`code/Raindrop.py:257-259`), `utils_rd.getStats` gets a numpy-2-safe
scalar max (`code/utils_rd.py:160` fails on numpy >= 1.24), numpy is seeded (the script only seeds
torch), `torch.optim.Adam` gets torch's own fused implementation where every parameter is a CUDA tensor (same algorithm, one or two
launches instead of a dozen), and `wandb` stays disabled.

The writer half (`write_dataset`) is validated against the reference's own loader in
`tests/test_compat_dataset.py`.  The end-to-end run with the HIP model needs BOTH a GPU and the reference
tree, which no single machine of the build environment has (GPU boxes receive only this repo); what IS
automated: `tests/test_compat_e2e.py` runs this runner end-to-end on CPU with the reference's own model
behind the shim (script plumbing, scheduler patch, dataset files, checkpoint save/load, all five splits),
and `tests/test_train_loop_gpu.py` drives the script's loop body (`code/Raindrop.py:290-324`) on the GPU
through `DeviceDataset` with the HIP model.
"""
import argparse
import os
import runpy
import sys

import numpy as np

from . import synth

DATASETS = {
    # name: (dir, record file, outcome file, split pattern, minutes?)
    "P12": ("P12data", "PTdict_list.npy", "arr_outcomes.npy", "phy12_split%d.npy"),
    "P19": ("P19data", "PT_dict_list_6.npy", "arr_outcomes_6.npy", "phy19_split%d_new.npy"),
    "PAM": ("PAMdata", "PTdict_list.npy", "arr_outcomes.npy", "PAM_split_%d.npy"),
}


def write_dataset(root, dataset, n_samples, seed=0):
    """Synthetic dataset in the reference's on-disk format.  Returns the dataset directory."""
    cfg = synth.make_config(dataset)
    ddir, rec_name, out_name, split_pat = DATASETS[dataset]
    base = os.path.join(root, ddir)
    os.makedirs(os.path.join(base, "processed_data"), exist_ok=True)
    os.makedirs(os.path.join(base, "splits"), exist_ok=True)
    rng = np.random.default_rng(seed)
    T, F = cfg["max_len"], cfg["d_inp"]
    b = synth.make_batch(cfg, n_samples, seed=seed)
    # raw (pre-normalisation) values: the reference treats "value > 0" as observed
    # (code/utils_rd.py:156,168), so observed entries are positive and missing ones exactly 0
    obs = b["src"][:, :, F:].numpy().transpose(1, 0, 2) > 0
    vals = np.where(obs, 1.0 + np.abs(b["src"][:, :, :F].numpy().transpose(1, 0, 2)), 0.0)   # [N,T,F]
    times = b["times"].numpy().T                                    # [N,T] hours
    if dataset == "PAM":
        # code/utils_rd.py:243-257 (tensorize_normalize_other): records are bare [T,F] arrays
        records = np.array([vals[i].astype(np.float64) for i in range(n_samples)], dtype=object)
        if records.ndim != 1:                                       # equal shapes collapse to 3-D: keep that form
            records = vals.astype(np.float64)
        # balanced labels: the script scores `roc_auc_score(one_hot(yval), ...)` (code/Raindrop.py:358), which needs every class
        # present in every validation split
        y = (rng.permutation(n_samples) % cfg["n_classes"]).reshape(n_samples, 1).astype(np.float64)
    else:
        records = np.empty(n_samples, dtype=object)
        for i in range(n_samples):
            static = rng.standard_normal(cfg["d_static"])
            records[i] = {"id": str(i), "static": static, "extended_static": static.copy(),
                          "arr": vals[i].astype(np.float64),
                          "time": (times[i] * 60.0).reshape(T, 1).astype(np.float64),   # minutes, utils_rd.py:235
                          "length": int((times[i] > 0).sum())}
        y = np.zeros((n_samples, 6 if dataset == "P12" else 1))
        y[:, -1] = (rng.random(n_samples) < 0.3).astype(np.float64)                     # ~30 % positives
    np.save(os.path.join(base, "processed_data", rec_name), records, allow_pickle=True)
    np.save(os.path.join(base, "processed_data", out_name), y, allow_pickle=True)
    for k in range(1, 6):
        perm = rng.permutation(n_samples)
        a, c = int(0.8 * n_samples), int(0.9 * n_samples)
        split = np.empty(3, dtype=object)
        split[0], split[1], split[2] = perm[:a], perm[a:c], perm[c:]
        np.save(os.path.join(base, "splits", split_pat % k), split, allow_pickle=True)
    return base


HIP_SHIM = ("# drop-in shim: the training script does `from models_rd import *`\n"
            "from raindrop_amd.models_rd import *  # noqa: F401,F403\n"
            "from raindrop_amd.models_rd import Raindrop_v2, PositionalEncodingTF  # noqa: F401\n")


def make_workspace(root, dataset, n_samples, reference, seed=0, model_shim=None):
    """`model_shim`: source text of `<ws>/code/models_rd.py` (default: re-export raindrop_amd.models_rd).  The CPU plumbing
    test passes a shim that re-exports the reference's own model instead (tests/test_compat_e2e.py)."""
    write_dataset(root, dataset, n_samples, seed)
    code = os.path.join(root, "code")
    os.makedirs(code, exist_ok=True)
    os.makedirs(os.path.join(root, "models"), exist_ok=True)
    for f in ("Raindrop.py", "utils_rd.py"):
        dst = os.path.join(code, f)
        if not os.path.lexists(dst):
            os.symlink(os.path.join(reference, "code", f), dst)
    with open(os.path.join(code, "models_rd.py"), "w") as fh:
        fh.write(HIP_SHIM if model_shim is None else model_shim)
    return code


def get_stats_numpy2(P_tensor):
    """`utils_rd.getStats` (code/utils_rd.py:149-161) with the one line numpy >= 1.24 rejects
    (`np.max([stdf[f], eps])` on a ragged list) written as a scalar max; same values."""
    N, T, F = P_tensor.shape
    Pf = P_tensor.transpose((2, 0, 1)).reshape(F, -1)
    mf, stdf = np.zeros((F, 1)), np.ones((F, 1))
    for f in range(F):
        v = Pf[f, :]
        v = v[v > 0]
        mf[f] = np.mean(v)
        stdf[f] = max(float(np.std(v)), 1e-7)
    return mf, stdf


def _compat_patches(seed):
    import torch
    np.random.seed(seed)
    torch.manual_seed(seed)
    import utils_rd                                  # the reference's, via the workspace symlink
    utils_rd.getStats = get_stats_numpy2
    orig = torch.optim.lr_scheduler.ReduceLROnPlateau.__init__

    def init(self, *a, verbose=None, **k):      # kwarg removed in torch >= 2.7 (code/Raindrop.py:259)
        return orig(self, *a, **k)
    torch.optim.lr_scheduler.ReduceLROnPlateau.__init__ = init
    # torch.optim.Adam(model.parameters(), lr=..) (code/Raindrop.py:256) picks the multi-tensor "foreach" implementation by default:
    # ~12 launches and 0.28 ms of host time per step over this model's 100+ parameter tensors, as much as the whole captured
    # forward + backward.  torch's own FUSED implementation (same algorithm, one or two launches) is selected instead when every
    # parameter is a CUDA tensor and the script did not choose an implementation itself (RD_COMPAT_FUSED_ADAM=0: leave the default).
    if os.environ.get("RD_COMPAT_FUSED_ADAM", "1") != "0":
        orig_adam = torch.optim.Adam.__init__

        def adam_init(self, params, *a, **k):
            params = list(params)
            flat = [q for p in params for q in (p["params"] if isinstance(p, dict) else [p])]
            if "fused" not in k and "foreach" not in k and flat and all(isinstance(q, torch.Tensor) and q.is_cuda and q.is_floating_point() for q in flat):
                k["fused"] = True
            return orig_adam(self, params, *a, **k)
        torch.optim.Adam.__init__ = adam_init


def run(root, dataset, n_samples, reference, seed=0, extra_argv=(), model_shim=None):
    """Execute the reference's `code/Raindrop.py` byte-for-byte in the workspace; returns the script's globals
    (`acc_arr`, `auprc_arr`, ... as the script leaves them)."""
    code = make_workspace(root, dataset, n_samples, reference, seed, model_shim)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    added = [code, repo]
    sys.path[:0] = added
    for m in ("models_rd", "utils_rd"):          # a previous run's (or the oracle's) modules of the same top-level name
        sys.modules.pop(m, None)
    _compat_patches(seed)
    cwd = os.getcwd()
    os.chdir(code)                               # the script uses '../P19data', '../models/' (Raindrop.py:74-86)
    argv = sys.argv
    sys.argv = ["Raindrop.py", "--dataset", dataset, "--splittype", "random"] + list(extra_argv)
    try:
        return runpy.run_path(os.path.join(code, "Raindrop.py"), run_name="__main__")
    finally:
        sys.argv = argv
        os.chdir(cwd)
        for a in added:
            if a in sys.path:
                sys.path.remove(a)
        for m in ("models_rd", "utils_rd"):
            sys.modules.pop(m, None)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="P19", choices=sorted(DATASETS))
    ap.add_argument("--samples", type=int, default=2000)
    ap.add_argument("--root", default="/tmp/raindrop_compat_ws")
    ap.add_argument("--reference", default=os.environ.get("RAINDROP_REFERENCE", "/root/reference"))
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    if not os.path.isfile(os.path.join(a.reference, "code", "Raindrop.py")):
        raise SystemExit("reference tree not found at %s" % a.reference)
    run(a.root, a.dataset, a.samples, a.reference, a.seed)
