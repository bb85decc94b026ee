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


# ---- what a green run covered -------------------------------------------------------------------------------------------
# Some GPU tests rotate their inputs from run to run (whole-frame seeds, oracle-window positions, the stress pool).  A red run
# prints its seed in the failure; a GREEN run used to leave no trace of what it had covered.  Every rotating value now goes
# through rotating_seed(): ONE value per session (or SRCNN_TEST_SEED to replay), recorded per use, written to
# gpurun_out/test_seeds.json when the session ends and echoed in pytest's last lines, so the tail of the log carries it.
_SESSION_SEED = int(os.environ.get("SRCNN_TEST_SEED", "0")) or (int(__import__("time").time()) & 0xFFFFF)
_SEEDS_USED = {}


def rotating_seed(what):
    _SEEDS_USED[what] = _SESSION_SEED
    return _SESSION_SEED


def pytest_sessionfinish(session, exitstatus):
    if not _SEEDS_USED:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "test_seeds.json"), "w") as f:
            json.dump({"session_seed": _SESSION_SEED, "replay": "SRCNN_TEST_SEED=%d" % _SESSION_SEED, "exitstatus": int(exitstatus),
                       "used_by": sorted(_SEEDS_USED)}, f, indent=1)
    except OSError:
        pass


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if _SEEDS_USED:
        terminalreporter.write_line("rotating inputs of this run: SRCNN_TEST_SEED=%d (%s) -> gpurun_out/test_seeds.json" %
                                    (_SESSION_SEED, ", ".join(sorted(_SEEDS_USED))))


@pytest.hookimpl(hookwrapper=True)
def pytest_pyfunc_call(pyfuncitem):
    """The suite also runs against the strict-only build (SRCNN_AMD_LIB=libsrcnn_amd/lib/strict/libsrcnn_amd.so), which refuses
    every non-parity mode with SRCNN_E_UNSUPPORTED: a test that asks for one is skipped there, not failed."""
    outcome = yield
    exc = outcome.excinfo
    if exc and getattr(exc[1], "code", None) == -203 and "strict-only build" in str(exc[1]):
        outcome.force_exception(pytest.skip.Exception("strict-only build: " + str(exc[1])))


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
