#!/usr/bin/env python3
"""Measured error of the non-parity tiers against the STRICT tier's output on the same frame.  Strict mode is
bit-identical to the reference (tests/test_gpu_parity.py), so this is max|dY| vs the reference without needing
the CPU checker, at any frame size.  Usage: tools/fast_error.py [h w]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import libsrcnn_amd as S
from libsrcnn_amd import synth

S.init(0)
shapes = [(160, 224), (67, 131), (1080, 1920)]
if len(sys.argv) == 3:
    shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for h, w in shapes:
    for kind in ("smooth", "noise"):
        y = synth.plane(h, w, synth.SEED0 + 3, kind)
        S.set_mode(S.MODE_STRICT)
        ref = S.y_upscale2x(y).astype(np.float64)
        for mode, name in ((S.MODE_FAST, "fast"), (S.MODE_FAST_F16, "fast_f16")):
            S.set_mode(mode)
            got = S.y_upscale2x(y).astype(np.float64)
            d = np.abs(got - ref)
            idx = np.unravel_index(int(np.argmax(d)), d.shape)
            print("%4dx%-4d %-7s %-9s max|dY| = %.3e at %s  mean|dY| = %.3e  frac(|dY|>1e-4) = %.4f" %
                  (w, h, kind, name, d.max(), idx, d.mean(), (d > 1e-4).mean()))
S.set_mode(S.MODE_STRICT)
