"""Every resampler the library can select for a plane up-scale gives the same bits (GPU).

The default is k_rs2d_dma (LDS-DMA patch prefetch); SRCNN_RS_DMA=0 selects k_rs2d<0>, SRCNN_RESAMPLE_2PASS=1 the generic
fused kernel and SRCNN_RESAMPLE_2PASS=1 the two separate passes (the form that is closest to src/frawscale.cpp:238-385).
The selection is read once at library load, hence one subprocess per selection; each prints a sha256 per case, and the
default's output of the first cases is also compared with the oracle in this process.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from libsrcnn_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (h, w, dh, dw, filter id): multi-tile widths and heights, widths that are not a multiple of 4 (scalar stores), a single
# column / row of blocks, ratios whose tables have 3 / 5 / 8-tap rows, and a ratio between 1 and 2 (wide LDS rows)
CASES = [
    (300, 700, 600, 1400, 2),
    (301, 517, 602, 1034, 2),
    (97, 131, 291, 393, 2),
    (64, 900, 128, 1801, 2),
    (211, 333, 316, 499, 3),
    (150, 260, 300, 520, 1),
    (150, 260, 600, 1040, 4),
    (40, 50, 83, 101, 0),
    (17, 1200, 34, 2400, 2),
    (1200, 17, 2400, 34, 2),
]

_CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import libsrcnn_amd as S
from libsrcnn_amd import synth
S.init(0)
cases = json.loads(sys.argv[2])
out = []
for i, (h, w, dh, dw, f) in enumerate(cases):
    y = synth.plane(h, w, 4242 + i, "noise" if i & 1 else "smooth")
    if i == 2:                                  # special values travel through every variant the same way
        y[3, 5] = np.inf; y[40, 100] = -np.inf; y[60, 7] = np.nan; y[0, 0] = -0.0
    r = S.resample(y, dw, dh, f)
    out.append(hashlib.sha256(r.tobytes()).hexdigest())
    if i == 0:
        out.append(hashlib.sha256(S.y_upscale2x(y).tobytes()).hexdigest())
print("HASHES " + json.dumps(out))
"""


def run_variant(env):
    r = subprocess.run([sys.executable, "-c", _CHILD, ROOT, json.dumps(CASES)], env=dict(os.environ, **env),
                       capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("HASHES ")]
    assert r.returncode == 0 and line, r.stdout[-400:] + r.stderr[-800:]
    return json.loads(line[0][7:])


def test_all_plane_resamplers_agree_bit_for_bit(srcnn, oracle_lib):
    base = run_variant({})
    for env in ({"SRCNN_RS_DMA": "0"}, {"SRCNN_RESAMPLE_2PASS": "1"}, {"SRCNN_RS_TPB": "1"},
                {"SRCNN_RS_TPB": "3", "SRCNN_RS_DMA": "0"}):
        got = run_variant(env)
        assert got == base, "%r differs from the default resampler in cases %r" % (
            env, [i for i, (a, b) in enumerate(zip(got, base)) if a != b])
    # and the default is the oracle's result (first two cases; the second has a width that is not a multiple of 4)
    for i in (0, 1):
        h, w, dh, dw, f = CASES[i]
        y = synth.plane(h, w, 4242 + i, "noise" if i & 1 else "smooth")
        want = oracle_lib.resample(y, dw, dh, f)
        got = srcnn.resample(y, dw, dh, f)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), CASES[i]
        idx = i if i == 0 else i + 1                 # case 0 contributes two hashes
        assert hashlib.sha256(got.tobytes()).hexdigest() == base[idx]
