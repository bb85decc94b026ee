#!/usr/bin/env python3
"""Randomized parity campaign on the GPU box (heavier than the unit tests): random shapes, data kinds, filters/ratios and
band splits of the float Y path, and -- round 3 -- random RGB / RGBA images, ratios and filters through srcnn_process_u8
(fused colour shell, the plane fallback for down-scales / identity axes, small and banded large images), strict mode vs the
oracle, bit for bit.  Exit code != 0 on any mismatch.

    python3 tools/campaign.py [n_float_cases] [n_image_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import libsrcnn_amd as S
from libsrcnn_amd import synth
import oracle

def main(n=150, seed=1234):
    if not os.environ.get("SRCNN_DEVICES"):
        S.init(0)                                   # SRCNN_DEVICES=0,0,0: the library self-initialises K (virtual) contexts
    o = oracle.Oracle()
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(n):
        h, w = int(rng.integers(1, 120)), int(rng.integers(1, 260))
        if rng.random() < 0.1:                          # a tenth of the cases span many column tiles / several row tiles
            h, w = int(rng.integers(100, 500)), int(rng.integers(300, 1500))
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
        if k % 50 == 49:
            print("  ... %d float cases done" % (k + 1), flush=True)
        if not ok:
            bad += 1
            d = np.abs(got.astype(np.float64) - want) if got.shape == want.shape else np.array([np.inf])
            print("MISMATCH case %d: %dx%d %s %s max|d|=%g" % (k, w, h, kind, what, float(np.nanmax(d))))
    print("campaign: %d float cases, %d mismatches" % (n, bad), flush=True)
    return bad


def images(n=60, seed=4321):
    """ProcessSRCNN surface: everything the colour shell can meet."""
    o = oracle.Oracle()
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(n):
        big = rng.random() < float(os.environ.get("CAMPAIGN_BIG", 0.25))   # above the 8 MB threshold: banded, pipelined, staged (dealt over the contexts when there are several)
        h = int(rng.integers(600, 1100)) if big else int(rng.integers(1, 200))
        w = int(rng.integers(900, 1500)) if big else int(rng.integers(1, 300))
        d = int(rng.choice([3, 4]))
        filt = int(rng.integers(0, 5))
        m = float(rng.choice([2.0, 2.0, 2.0, 1.5, 3.0, 2.5, 1.25, 0.75, 0.5, 4.0])) if not big else float(rng.choice([2.0, 2.0, 1.5, 1.7]))
        if int(np.float32(w) * np.float32(m)) < 1 or int(np.float32(h) * np.float32(m)) < 1:
            continue
        if h * w * m * m * d > 60e6:
            m = 2.0
        img = rng.integers(0, 256, (h, w, d), dtype=np.uint8)
        if rng.random() < 0.3:                                      # flat regions and saturated extremes
            img[: h // 2] = rng.choice([0, 255, 16, 128])
        want_rgb, want_conv = o.process(img, m, filt)
        got_rgb, got_conv = S.process_u8(img, m, filt, want_conv=True)
        ok = got_rgb.shape == want_rgb.shape and np.array_equal(got_rgb, want_rgb) and np.array_equal(got_conv, want_conv)
        if not ok:
            bad += 1
            print("MISMATCH image case %d: %dx%dx%d x%.2f filter %d" % (k, w, h, d, m, filt))
    print("campaign: %d image cases, %d mismatches" % (n, bad))
    return bad


if __name__ == "__main__":
    nf = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    ni = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1234
    b = main(nf, seed) + images(ni, seed + 1)
    sys.exit(1 if b else 0)
