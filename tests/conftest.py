"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` covers the oracle against the golden vectors, the host logic, and that the C-ABI library
loads and exports every declared symbol; `-m gpu` holds the parity tests proper (HIP path vs oracle).
"""
import json
import os
import sys

import pytest

# The multi-device path is rehearsed on ONE GPU (every shard on device 0): its persistent decision kernels wait for each
# other, so each needs a hardware queue of its own.  The HIP runtime maps streams onto 4 hardware queues by default;
# this must be set before the runtime initialises (i.e. before torch / liblpx touch the GPU).  On a real multi-GPU
# node every device has its own queues and nothing needs to be set.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / at round end)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """Some GPU tests hand torch CUDA tensors / streams to liblpx.  torch ships its own copy of the HIP runtime; when
    it is initialised late, after liblpx has worked the GPU for a while through the system runtime, its device
    discovery has been seen to fail on the test boxes ("no ROCm-capable device is detected").  Initialising it first
    is harmless on a machine without a GPU (is_available() is False there)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def reference_vectors():
    with open(os.path.join(GOLDEN, "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def decimal_goldens():
    with open(os.path.join(GOLDEN, "decimal_goldens.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


# ---- the two arithmetic modes of the HIP path ---------------------------------------------------------------------
# Every GPU parity test runs twice: "plain" (the library's default: product and difference of every update rounded
# separately, checked against the oracle's fp64 instantiation) and "fused" (LPX_OPT_FUSED: fused multiply-add updates,
# checked against the oracle's fused instantiation).  The GPU test modules take their `lps` / `oracle` fixtures from the
# two helpers below, so a test says `oracle.FP64` and gets the checker of the mode it runs in.
@pytest.fixture(scope="session", params=["plain", "fused"])
def arith(request):
    return request.param


class ArithOracle:
    """oracle.pyoracle with FP64 standing for the binary instantiation of the current arithmetic mode."""

    def __init__(self, mod, mode):
        self._mod = mod
        self.mode = mode
        self.FP64 = mod.FP64_FUSED if mode == "fused" else mod.FP64

    def __getattr__(self, name):
        return getattr(self._mod, name)


def package_in_mode(mode):
    """The host package with new handles defaulting to `mode` (set_default_arithmetic)."""
    import linear_programming_solver_amd as pkg
    from linear_programming_solver_amd import _lib
    _lib.lib()
    assert _lib.lib().lpx_device_count() >= 1, "no HIP device visible"
    pkg.set_default_arithmetic(mode)
    return pkg
