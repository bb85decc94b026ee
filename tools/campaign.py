#!/usr/bin/env python3
"""One-off randomized parity campaign on the GPU box (heavier than the unit tests): random shapes, data kinds,
filters/ratios and band splits, strict mode vs the oracle, bit for bit.  Exit code != 0 on any mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import libsrcnn_amd as S
from libsrcnn_amd import synth
import oracle

def main(n=150, seed=1234):
    S.init(0)
    o = oracle.Oracle()
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(n):
        h, w = int(rng.integers(1, 120)), int(rng.integers(1, 260))
        kind = "noise" if rng.random() < 0.5 else "smooth"
        y = synth.plane(h, w, int(rng.integers(0, 1 << 30)), kind)
        if rng.random() < 0.15:
            y *= np.float32(rng.choice([0.0, 1e-3, 4.0, -1.0]))
        mode = rng.integers(0, 3)
        if mode == 0:                                   # 2x whole frame
            got, want, what = S.y_upscale2x(y), o.y_path(y), "2x"
        elif mode == 1:                                 # band of the 2x frame
            row0 = int(rng.integers(0, 2 * h)); rows = int(rng.integers(1, 2 * h - row0 + 1))
            got, want, what = S.y_upscale2x_band(y, row0, rows), o.y_path(y)[row0:row0 + rows], "band %d+%d" % (row0, rows)
        else:                                           # general path
            filt = int(rng.integers(0, 5))
            dw = max(1, int(w * rng.uniform(0.5, 3.2))); dh = max(1, int(h * rng.uniform(0.5, 3.2)))
            if dw == w and dh == h:
                dw += 1
            got, want, what = S.y_path(y, dw, dh, filt), o.y_path(y, dw, dh, filt), "filter %d -> %dx%d" % (filt, dw, dh)
        ok = got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
        if not ok:
            bad += 1
            d = np.abs(got.astype(np.float64) - want) if got.shape == want.shape else np.array([np.inf])
            print("MISMATCH case %d: %dx%d %s %s max|d|=%g" % (k, w, h, kind, what, float(np.nanmax(d))))
    print("campaign: %d cases, %d mismatches" % (n, bad))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 150))
