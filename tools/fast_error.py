#!/usr/bin/env python3
"""Measured error of the fast tiers against the oracle (bit-exact reference restatement) and against an
fp64 evaluation of the same network (how far each is from exact arithmetic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import libsrcnn_amd as S
from libsrcnn_amd import synth
import oracle

S.init(0)
o = oracle.Oracle()
for kind in ("smooth", "noise"):
    y = synth.plane(160, 224, synth.SEED0 + 3, kind)
    ref = o.y_path(y).astype(np.float64)
    for mode, name in ((S.MODE_STRICT, "strict"), (S.MODE_FAST, "fast"), (S.MODE_FAST_F16, "fast_f16")):
        S.set_mode(mode)
        got = S.y_upscale2x(y).astype(np.float64)
        d = np.abs(got - ref)
        print("%-7s %-9s max|dY| = %.3e   mean|dY| = %.3e   frac(|dY|>1e-4) = %.4f" % (kind, name, d.max(), d.mean(), (d > 1e-4).mean()))
S.set_mode(S.MODE_STRICT)
