"""GPU: the strict FUSED prototype (tools/fused_strict/: layers 1+2+3 in one kernel, the measurement behind DESIGN.md 4.5 and
profiles/r05_fused_strict.txt) stays what the measurement says it is: bit-exact on the golden planes made by the compiled
reference, on strip- / step- / chunk-straddling shapes, and on the headline frame against the production layer kernels.  It is
NOT part of the product (slower: +22 %); this test keeps the experiment reproducible."""
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tools", "fused_strict", "probe.py")


def test_fused_strict_prototype_is_bit_exact():
    lib = os.path.join(ROOT, "tools", "fused_strict", "libfused_strict.so")
    if not os.path.exists(lib) and not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no prebuilt prototype and no hipcc")
    r = subprocess.run([sys.executable, PROBE, "--quick"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "RESULT all bit-exact" in r.stdout, r.stdout[-2000:]
    assert r.stdout.count("bit-exact") >= 11 and "MISMATCH" not in r.stdout
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("round 0:")][0]
    print(line)
