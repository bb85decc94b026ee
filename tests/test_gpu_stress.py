"""Stress of the LDS-DMA staged layer kernels (k_conv12_mfma, k_conv3 weight image) and of the staging they
replaced: thousands of small-shape parity launches interleaved with host-to-device copies, in child processes whose stderr
is kept (gpurun_out/stress_*.err) -- a runtime abort inside pytest's own process loses its message to the capture buffer,
which is how the one abort seen in round 3 came to be "without any message".  Every result is the oracle's, bit for bit
(/root/reference/src/libsrcnn.cpp:350-529 via oracle/)."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "stress_worker.py")
OUT = os.path.join(ROOT, "gpurun_out")

CASES = [
    ("dma-default", {}, 12000),
    ("dma-no-quarter-spread", {"SRCNN_CONV12_SPREAD": "0"}, 4000),
    ("no-dma-staging", {"SRCNN_CONV12_DMA": "0", "SRCNN_CONV3_WDMA": "0"}, 4000),
]
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rotating_seed          # noqa: E402


def stress_seed():
    return rotating_seed("stress pool of small shapes")


@pytest.fixture(scope="module")
def pool_file(tmp_path_factory):
    """The cases and their oracle answers, computed ONCE (the oracle is the slow part) and shared by the children."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stress_worker
    path = str(tmp_path_factory.mktemp("stress") / "pool.npz")
    stress_worker.save_pool(path, 40, stress_seed())
    return path


@pytest.mark.parametrize("name,env,iters", CASES, ids=[c[0] for c in CASES])
def test_small_shape_launches_interleaved_with_h2d_copies(name, env, iters, pool_file):
    seed = stress_seed()
    e = dict(os.environ)
    e.pop("SRCNN_DEVICES", None)
    e.update(env)
    e.setdefault("AMD_LOG_LEVEL", "1")            # runtime errors (queue faults, aborted packets) are printed
    os.makedirs(OUT, exist_ok=True)
    err_path = os.path.join(OUT, "stress_%s.err" % name)
    with open(err_path, "w") as err:
        r = subprocess.run([sys.executable, WORKER, str(iters), str(seed), "40", pool_file], env=e, stdout=subprocess.PIPE,
                           stderr=err, text=True, timeout=600)
    tail = open(err_path).read()[-3000:]
    assert r.returncode == 0, "stress '%s' seed %d: exit %d\nstdout: %s\nstderr tail (kept in %s):\n%s" % (
        name, seed, r.returncode, r.stdout[-1000:], err_path, tail)
    res = json.loads(r.stdout.strip().splitlines()[-1])
    print("stress", name, "seed", seed, res)
    assert res["mismatches"] == 0 and res["iterations"] == iters, res
    with open(os.path.join(OUT, "stress_%s.json" % name), "w") as f:
        json.dump(dict(res, seed=seed), f)
