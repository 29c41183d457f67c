"""pytest configuration: the `gpu` marker (tests that need a real MI355X) and shared helpers."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The CPU suite is thousands of tiny tensor ops (per-sample / per-edge oracle loops).  With the default one thread
# per core their OpenMP fork/join dominates, and on a busy or virtualised host it degrades by 10-100x (measured
# here: 2000 [34x240]@[240x240] products take 1.0 s on 1 thread, 25 s on 4, 132 s on 8 while a neighbour is
# active).  One thread is both faster and predictable; RD_TEST_THREADS overrides.
_NT = str(max(1, int(os.environ.get("RD_TEST_THREADS", "1"))))
os.environ.setdefault("OMP_NUM_THREADS", _NT)    # inherited by the world-size-2 gloo worker processes
try:
    import torch
    torch.set_num_threads(int(_NT))
except Exception:                                # torch-free collection (e.g. symbol tests only)
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / the round-end driver)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _eager_operator_surface(request, monkeypatch):
    """The parity tests of this suite exercise the operator-by-operator surface of the module (one C-ABI call per operator under
    autograd); since round 5 a training call takes the captured step by default (raindrop_amd/graph_module.py), which
    tests/test_graph_module_gpu.py covers -- that file and tests/test_trajectory_gpu.py manage the switch themselves, the golden /
    reference-parity tests of tests/test_gpu_parity.py run BOTH ways (model.graph_step False / True), every other test with it off."""
    if "test_graph_module_gpu" not in request.node.nodeid and "test_trajectory_gpu" not in request.node.nodeid:
        monkeypatch.setenv("RD_MODULE_GRAPH", "0")
    yield
