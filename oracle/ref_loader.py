"""TEST INFRASTRUCTURE ONLY (oracle O1) -- run the reference's own files, unmodified, on CPU.

Only `tests/`, `tests/golden/make_goldens.py`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this module; nothing under `raindrop_amd/` does.

What it does (strategy O1):
  * puts `oracle/pyg_shim` (plain-torch stand-ins for torch_geometric / torch_scatter /
    torch_sparse, none of which is installed or pinned by the reference) and
    `<reference>/code` on `sys.path`;
  * neutralises the four things that stop `code/models_rd.py` importing on Linux / CPU / torch 2.10:
      - `os.add_dll_directory(...)`            (`code/models_rd.py:8-9`, Windows only)
      - hard-coded `.cuda()`                   (`code/models_rd.py:42,239,241,299,307,315,321`)
      - `adj[torch.eye(n).byte()] = 1`         (`code/models_rd.py:308`; uint8 masks rejected by torch>=2)
  * imports `models_rd`, `Ob_propagation`, `transformer_conv` from the reference tree.

All arithmetic in `forward()/message()` is then the reference's own code; only PyG's
gather / softmax / scatter / glorot are restated (in the shim).  The reference tree is read
from `RAINDROP_REFERENCE` or `/root/reference`; it does NOT exist on the GPU box, so callers
must treat `available() == False` as "skip", never as a failure of the product.
"""
import contextlib
import importlib
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_SHIM = os.path.join(_HERE, "pyg_shim")
_MODS = ("models_rd", "Ob_propagation", "transformer_conv")


def reference_root():
    return os.environ.get("RAINDROP_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(reference_root(), "code", "models_rd.py"))


@contextlib.contextmanager
def _patched():
    """Monkeypatches live only while reference code is being imported or executed."""
    import torch
    import torch.nn as nn

    saved = dict(
        add_dll=getattr(os, "add_dll_directory", None),
        t_cuda=torch.Tensor.cuda, m_cuda=nn.Module.cuda, t_byte=torch.Tensor.byte,
    )
    os.add_dll_directory = lambda p: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.byte = lambda self: self.bool()
    try:
        yield
    finally:
        if saved["add_dll"] is None:
            del os.add_dll_directory
        else:
            os.add_dll_directory = saved["add_dll"]
        torch.Tensor.cuda = saved["t_cuda"]
        nn.Module.cuda = saved["m_cuda"]
        torch.Tensor.byte = saved["t_byte"]


class _Ref:
    """Handle on the imported reference modules; `run()` executes a callable under the patches."""

    def __init__(self, mods):
        self.models_rd, self.Ob_propagation, self.transformer_conv = mods

    @staticmethod
    def run(fn, *a, **k):
        with _patched():
            return fn(*a, **k)


_cached = None


def load():
    """Import the reference modules (once).  Raises FileNotFoundError if the tree is absent."""
    global _cached
    if _cached is not None:
        return _cached
    if not available():
        raise FileNotFoundError("reference tree not found at %s" % reference_root())
    code = os.path.join(reference_root(), "code")
    # our own package also ships modules called models_rd / Ob_propagation / transformer_conv
    # (inside raindrop_amd/); make sure the top-level names resolve to the reference here.
    for m in _MODS:
        if m in sys.modules and not getattr(sys.modules[m], "__file__", "").startswith(code):
            del sys.modules[m]
    sys.path.insert(0, code)
    sys.path.insert(0, _SHIM)
    try:
        with _patched():
            mods = tuple(importlib.import_module(m) for m in _MODS)
    finally:
        sys.path.remove(code)
        sys.path.remove(_SHIM)
    _cached = _Ref(mods)
    return _cached


def load_models_rd_with_beta():
    """The reference's `code/models_rd.py` with ONE literal flipped in memory: `use_beta = False` -> `use_beta = True`
    (`code/models_rd.py:317`, the switch the reference hard-wires off).  Everything else is the file as it lies under the
    reference tree; nothing is written anywhere.  Used only to generate / re-check the `*_beta` golden fixtures."""
    import types
    load()                                           # Ob_propagation / transformer_conv imported under the shim
    path = os.path.join(reference_root(), "code", "models_rd.py")
    with open(path) as fh:
        text = fh.read()
    assert text.count("use_beta = False") == 1, "the reference's use_beta literal moved"
    text = text.replace("use_beta = False", "use_beta = True")
    mod = types.ModuleType("models_rd_use_beta")
    mod.__file__ = path
    code = os.path.join(reference_root(), "code")
    sys.path.insert(0, code)
    sys.path.insert(0, _SHIM)
    try:
        with _patched():
            exec(compile(text, path, "exec"), mod.__dict__)
    finally:
        sys.path.remove(code)
        sys.path.remove(_SHIM)
    return mod


def build_raindrop_v2(cfg, global_structure, sensor_wise_mask=False, use_beta=False):
    """Construct the reference `Raindrop_v2` exactly as `code/Raindrop.py:245-251` does."""
    ref = load()
    args = (cfg["d_inp"], cfg["d_model"], cfg["nhead"], cfg["nhid"], cfg["nlayers"], cfg["dropout"],
            cfg["max_len"], cfg["d_static"], cfg["MAX"], 0.5, cfg["aggreg"], cfg["n_classes"],
            global_structure)
    kw = dict(sensor_wise_mask=sensor_wise_mask)
    if not cfg["static"]:
        kw["static"] = False
    cls = load_models_rd_with_beta().Raindrop_v2 if use_beta else ref.models_rd.Raindrop_v2
    return ref.run(cls, *args, **kw)


def forward(model, src, static, times, lengths):
    """`model.forward(P, Pstatic, Ptime, lengths)` (`code/Raindrop.py:319`) under the CPU patches."""
    return load().run(model.forward, src, static, times, lengths)
