"""CPU: the C-ABI library builds, loads, and exports exactly what include/raindrop_hip.h declares
(no compute calls -- there is no GPU here)."""
import os
import re

import pytest

from raindrop_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "raindrop_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rd_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from raindrop_amd import build
        build.build(verbose=False)
    return _lib.load()


def test_header_and_binding_agree():
    assert _declared() == sorted(_lib.SIGNATURES)


def test_every_declared_symbol_is_exported(lib):
    for name in _declared():
        assert hasattr(lib, name), name


def test_no_undeclared_exports(lib):
    """Everything the shared library exports under the rd_ prefix is declared: the ABI in include/raindrop_hip.h, the profiling
    hooks (no binding, tools only) in include/raindrop_hip_debug.h."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("rd_")})
    text = open(os.path.join(ROOT, "include", "raindrop_hip_debug.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    debug = sorted(set(re.findall(r"\b(rd_debug_[a-z0-9_]+)\s*\(", text)))
    assert all(n.startswith("rd_debug_") for n in debug)
    assert exported == sorted(_declared() + debug)


def test_identity(lib):
    assert lib.rd_version() == 1
    assert lib.rd_arch() == b"gfx950"


def test_argument_errors_are_reported_before_launch(lib):
    # NULL tensors / bad dims must come back as RD_EINVAL with a message, without touching a GPU
    rc = lib.rd_linear_fwd(4, 0, 8, None, 8, None, None, None, 8, 0, None)
    assert rc == -1 and b"bad dims" in lib.rd_last_error()
    rc = lib.rd_edge_softmax(0, None, None, None, None)
    assert rc == -1
    shp = _lib.shape(2, 5, 3, 4)
    import ctypes
    assert lib.rd_msgpass_workspace_bytes(ctypes.byref(shp)) > 0
    rc = lib.rd_msgpass_fwd(ctypes.byref(shp), *([None] * 7), 0.0, 0, None, 12, None, 0, None)
    assert rc == -1 and b"NULL" in lib.rd_last_error()
    # batch feed: bad dims, NULL tensors; an empty batch is a no-op that needs no device
    assert lib.rd_batch_gather(5, 2, 0, 0, 10, *([None] * 12)) == -1 and b"bad dims" in lib.rd_last_error()
    assert lib.rd_batch_gather(5, 2, 6, 0, 10, *([None] * 12)) == -1 and b"NULL" in lib.rd_last_error()
    assert lib.rd_batch_gather(5, 0, 6, 0, 10, *([None] * 12)) == 0
    # weight-gradient workspace query is pure host arithmetic and grows with the reduction length
    assert lib.rd_linear_bwd_weight_workspace_bytes(15360, 152, 272) >= lib.rd_linear_bwd_weight_workspace_bytes(256, 152, 272) > 0


def test_product_path_fails_loudly_without_device():
    import torch
    from raindrop_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.RaindropHipError):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4), torch.zeros(3))
