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
