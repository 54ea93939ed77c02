"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` covers the oracle against the golden vectors, the host logic, and that the C-ABI library
loads and exports every declared symbol; `-m gpu` holds the parity tests proper (HIP path vs oracle).
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun / at round end)")


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
