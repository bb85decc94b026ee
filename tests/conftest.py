import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def golden():
    class G:
        butterfly = np.load(os.path.join(GOLDEN, "butterfly.npz"))
        planes = np.load(os.path.join(GOLDEN, "y_planes.npz"))
        process = np.load(os.path.join(GOLDEN, "process_cases.npz"))
        resample = np.load(os.path.join(GOLDEN, "resample.npz"))
        known = json.load(open(os.path.join(GOLDEN, "known_answers.json")))
        weights = np.fromfile(os.path.join(GOLDEN, "weights_f32.bin"), dtype="<f4")
    return G


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    return oracle.Oracle()


@pytest.fixture(scope="session")
def srcnn():
    """The product binding, initialised on device 0.  Only GPU tests request this."""
    import libsrcnn_amd as S
    S.init(0)
    return S


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(got, want, what=""):
    got = np.ascontiguousarray(got, np.float32)
    want = np.ascontiguousarray(want, np.float32)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if not np.array_equal(bits(got), bits(want)):
        diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
        bad = int(np.count_nonzero(bits(got) != bits(want)))
        idx = np.unravel_index(int(np.argmax(diff)), diff.shape)
        raise AssertionError("%s: %d/%d elements differ, max|d|=%.3e at %s (got %r want %r)" %
                             (what, bad, got.size, float(diff.max()), idx, got[idx], want[idx]))
