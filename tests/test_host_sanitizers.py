"""CPU: the host code of the drop-in under AddressSanitizer+UBSan and ThreadSanitizer (make asan / make tsan).

SURVEY.md section 5: the reference ships without any sanitizer run and has latent out-of-bounds bugs in exactly the code
this project restates on the host (src/frawscale.cpp:185-193,249).  tests/host/host_sanitize.cpp drives the product's
contribution-table builder (csrc/resample_table.hpp), the ProcessSRCNN / ConfigureFilterSRCNN control flow
(csrc/dropin.cpp: argument checks, step-scaling loop, new[] ownership) and the oracle; the one device call
(srcnn_process_u8) is replaced, in that test binary only, by a stand-in that forwards to the oracle.
GPU-side guard bands are in tests/test_gpu_configs.py (no GPU sanitizer exists on this pool)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("target", ["asan", "tsan"])
def test_host_code_under_sanitizer(target):
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no host toolchain")
    r = subprocess.run(["make", "-s", "-C", ROOT, target], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and ("cannot find -l%s" % target in r.stderr or "lib%s" % target in r.stderr and "No such file" in r.stderr):
        pytest.skip("sanitizer runtime not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "all checks passed" in r.stdout
