#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (dev container only).

Everything written here is data: inputs and the outputs the compiled reference
(oracle/_ref/libsrcnn_ref.so, built by oracle/Makefile from /root/reference/src) produced for
them.  The butterfly pair is the reference's own published sample (Pictures/butterfly.png ->
butterfly_srcnn.png / butterfly_srcnn_convolution.png, README.md:38-53); we store the decoded
pixels and check that the compiled reference reproduces the published PNGs bit for bit before
writing anything.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
PICS = "/root/reference/Pictures"


def synth_plane(h, w, seed, kind="noise"):
    """Counter-based synthetic Y plane in [0,255] (same generator as libsrcnn_amd.synth)."""
    idx = np.arange(h * w, dtype=np.uint64).reshape(h, w)
    x = (idx + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(33); x = (x * np.uint64(0xFF51AFD7ED558CCD)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(33); x = (x * np.uint64(0xC4CEB9FE1A85EC53)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x ^= x >> np.uint64(33)
    u = (x >> np.uint64(40)).astype(np.float64) * 2.0 ** -24
    if kind == "noise":
        return (255.0 * u).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    s = 0.5 + 0.22 * np.sin(xx * 0.11 + 0.3 * seed) + 0.18 * np.cos(yy * 0.07 - xx * 0.013) + 0.08 * (u - 0.5)
    return (255.0 * np.clip(s, 0, 1)).astype(np.float32)


def main():
    from PIL import Image
    oracle.build()
    ref = oracle.Reference()
    os.makedirs(GOLD, exist_ok=True)
    meta = {}

    # --- 1. the reference's own golden pair --------------------------------------------------
    rgb = np.array(Image.open(os.path.join(PICS, "butterfly.png")).convert("RGB"))
    want_rgb = np.array(Image.open(os.path.join(PICS, "butterfly_srcnn.png")).convert("RGB"))
    want_y = np.array(Image.open(os.path.join(PICS, "butterfly_srcnn_convolution.png")))
    got_rgb, got_y = ref.process(rgb, 2.0, oracle.FILTER_BICUBIC)
    assert np.array_equal(got_rgb, want_rgb) and np.array_equal(got_y, want_y), "reference != published PNGs"
    np.savez_compressed(os.path.join(GOLD, "butterfly.npz"), rgb_in=rgb, rgb_out=want_rgb, conv_y=want_y)
    meta["butterfly"] = {"rgb_out_sha256": hashlib.sha256(want_rgb.tobytes()).hexdigest(),
                         "conv_y_sha256": hashlib.sha256(want_y.tobytes()).hexdigest()}
    # RGBA and a non-2x / other-filter end-to-end case (small crops)
    crop = rgb[96:136, 80:128]
    rgba = np.dstack([crop, (np.arange(crop.shape[0] * crop.shape[1]).reshape(crop.shape[:2]) * 7 % 256).astype(np.uint8)])
    e2e = {"rgba_in": rgba}
    e2e["rgba_out"], e2e["rgba_conv"] = ref.process(rgba, 2.0, oracle.FILTER_BICUBIC)
    for name, filt in (("nearest", 0), ("bilinear", 1), ("lanczos3", 3), ("bspline", 4)):
        e2e["rgb_%s_out" % name], e2e["rgb_%s_conv" % name] = ref.process(crop, 2.0, filt)
    e2e["rgb_in"] = crop
    e2e["rgb_x15_out"], e2e["rgb_x15_conv"] = ref.process(crop, 1.5, oracle.FILTER_BICUBIC)
    e2e["rgb_x3_out"], e2e["rgb_x3_conv"] = ref.process(crop, 3.0, oracle.FILTER_BICUBIC)
    e2e["rgb_x4step_out"], e2e["rgb_x4step_conv"] = ref.process(crop[:20, :24], 4.0, oracle.FILTER_BICUBIC, step=True)
    e2e["rgb_x3step_out"], e2e["rgb_x3step_conv"] = ref.process(crop[:20, :24], 3.0, oracle.FILTER_BICUBIC, step=True)
    np.savez_compressed(os.path.join(GOLD, "process_cases.npz"), **e2e)

    # --- 2. float Y planes through the stage-level path ---------------------------------------
    planes = {}
    cases = [("noise_24x40", 24, 40, 1, "noise"), ("noise_29x37", 29, 37, 2, "noise"),
             ("smooth_33x65", 33, 65, 3, "smooth"), ("row_1x17", 1, 17, 4, "noise"),
             ("col_13x1", 13, 1, 5, "noise"), ("tiny_2x3", 2, 3, 6, "noise"), ("one_1x1", 1, 1, 7, "noise"),
             ("noise_70x9", 70, 9, 8, "noise"), ("smooth_7x130", 7, 130, 9, "smooth")]
    for name, h, w, seed, kind in cases:
        y = synth_plane(h, w, seed, kind)
        out, up, c1, c2 = ref.y_path(y, taps=True)
        planes[name + "_in"] = y
        planes[name + "_up"] = up
        planes[name + "_out"] = out
        if name == "noise_24x40":
            planes[name + "_c1"] = c1
            planes[name + "_c2"] = c2
    # out-of-range / special values survive the path identically (negatives, >255, tiny, zero)
    y = synth_plane(12, 16, 11, "noise")
    y[0, :4] = [-37.5, 300.25, 1e-30, 0.0]
    y[5, 5] = -0.0
    planes["wild_12x16_in"] = y
    planes["wild_12x16_out"], planes["wild_12x16_up"], _, _ = ref.y_path(y, taps=True)
    np.savez_compressed(os.path.join(GOLD, "y_planes.npz"), **planes)

    # --- 3. constant planes: portable known answers -------------------------------------------
    consts = {}
    for c in (0.0, 1.0, 64.0, 128.0, 200.0, 255.0):
        out = ref.y_path(np.full((12, 16), c, np.float32))
        assert np.all(out == out[0, 0])
        consts[repr(c)] = {"value": float(out[0, 0]), "bits": int(out.view(np.uint32)[0, 0])}
    meta["constant_planes"] = consts

    # --- 4. resampler alone: other filters and ratios -----------------------------------------
    rs = {}
    y = synth_plane(19, 23, 21, "smooth")
    rs["in"] = y
    for name, filt in (("nearest", 0), ("bilinear", 1), ("bicubic", 2), ("lanczos3", 3), ("bspline", 4)):
        rs[name + "_x2"] = ref.resample(y, 46, 38, filt)
        rs[name + "_x1p5"] = ref.resample(y, 34, 28, filt)
        rs[name + "_x3"] = ref.resample(y, 69, 57, filt)
        rs[name + "_down"] = ref.resample(y, 11, 9, filt)
    np.savez_compressed(os.path.join(GOLD, "resample.npz"), **rs)

    with open(os.path.join(GOLD, "known_answers.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for fn in sorted(os.listdir(GOLD)):
        print("%9d  %s" % (os.path.getsize(os.path.join(GOLD, fn)), fn))


if __name__ == "__main__":
    main()
