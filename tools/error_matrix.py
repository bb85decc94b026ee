#!/usr/bin/env python3
"""Per-layer error matrix: on whole 3840x2160 -> 7680x4320 frames of both generators and N seeds, the
max|dY| of every combination of {strict, relaxed} per layer against the STRICT output of the same frame -- which is the
reference's, bit for bit (tests/test_gpu_parity.py) -- together with the device time of each combination.  Relaxed means
(include/srcnn_amd.h SRCNN_RELAX_*, src/libsrcnn.cpp:395-410 / :433-437 / :500-517):
    L1   layer 1 as an FMA chain on the fp32 MFMA (one rounding per tap instead of two)
    L2   layer 2 likewise
    L3x  layer 3 with exact products (v_fma_f64 on widened operands), sums as the reference's
    L3f  layer 3 as fp32 FMA chains
plus the two existing non-parity tiers (SRCNN_MODE_FAST = L1+L2+L3f, SRCNN_MODE_FAST_F16).

    python3 tools/error_matrix.py [seeds=16] [h=2160] [w=3840]  > profiles/r04_error_matrix.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import libsrcnn_amd as S
from libsrcnn_amd import synth

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
h = int(sys.argv[2]) if len(sys.argv) > 2 else 2160
w = int(sys.argv[3]) if len(sys.argv) > 3 else 3840
S.init(0)
L = S.lib()
NAMES = {1: "L1", 2: "L2", 4: "L3x", 8: "L3f"}
COMBOS = [1, 2, 4, 8, 1 | 2, 1 | 4, 2 | 4, 1 | 2 | 4, 1 | 2 | 8]


def label(mask):
    return "+".join(n for b, n in NAMES.items() if mask & b)


def run(d_in, d_out, reps=1):
    for _ in range(reps):
        S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_out.ptr, None))
    S.sync()


def timed(d_in, d_out, reps=8):
    run(d_in, d_out, 2)
    a, b = S.Event(), S.Event()
    a.record()
    for _ in range(reps):
        S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_out.ptr, None))
    b.record()
    return a.elapsed_ms(b) / reps


rows = {}           # label -> list of (max, mean, differ)
times = {}
d_out = S.DeviceBuffer(4 * h * w * 4)
t0 = time.time()
print("# per-layer error matrix: %d seeds x {smooth, noise}, %dx%d -> %dx%d, device %s" % (nseeds, w, h, 2 * w, 2 * h, S.device_name()))
print("# every figure is against the STRICT output of the same frame (= the reference, bit for bit)")
for si in range(nseeds):
    for kind in ("smooth", "noise"):
        y = synth.plane(h, w, synth.SEED0 + 100 + si, kind)
        d_in = S.DeviceBuffer.from_numpy(y)
        S.set_mode(S.MODE_STRICT)
        run(d_in, d_out)
        ref = d_out.to_numpy(np.float32, (2 * h, 2 * w))
        if si == 0 and kind == "smooth":
            times["strict"] = timed(d_in, d_out)
        todo = [(label(m), S.MODE_RELAXED, m) for m in COMBOS] + [("MODE_FAST", S.MODE_FAST, 0), ("MODE_FAST_F16", S.MODE_FAST_F16, 0)]
        for name, mode, mask in todo:
            if mode == S.MODE_RELAXED:
                S.set_relaxation(mask)
            S.set_mode(mode)
            run(d_in, d_out)
            got = d_out.to_numpy(np.float32, (2 * h, 2 * w))
            d = np.abs(got.astype(np.float64) - ref)
            rows.setdefault(name, []).append((float(d.max()), float(d.mean()), float((d != 0).mean()), float((d > 1e-4).mean())))
            if si == 0 and kind == "smooth":
                times[name] = timed(d_in, d_out)
        d_in.free()
        print("#   seed %d %s done (%.0f s)" % (si, kind, time.time() - t0), flush=True)
S.set_mode(S.MODE_STRICT)
npx = 4 * h * w
print()
print("%-14s %10s %10s %10s %10s %12s %9s %9s" % ("relaxed", "max|dY|", "worst-mean", "differ", ">1e-4", "frames>7e-5", "ms/frame", "GPix/s"))
print("%-14s %10s %10s %10s %10s %12s %9.3f %9.2f" % ("(strict)", "0", "0", "0", "0", "0/%d" % (2 * nseeds), times["strict"], npx / times["strict"] / 1e6))
for name, v in rows.items():
    v = np.array(v)
    print("%-14s %10.3e %10.3e %10.4f %10.2e %12s %9.3f %9.2f" % (
        name, v[:, 0].max(), v[:, 1].max(), v[:, 2].max(), v[:, 3].max(), "%d/%d" % (int((v[:, 0] > 7e-5).sum()), len(v)),
        times[name], npx / times[name] / 1e6))
print()
print("# columns: max over all frames of max|dY|; largest per-frame mean|dY|; largest fraction of samples that differ at all;")
print("# largest fraction of samples off by more than 1e-4; frames whose max exceeds 7e-5 (the margin under the 1e-4 bar);")
print("# device time per frame (resampler + layers, 8 frames back to back, first seed) and the rate that corresponds to.")
