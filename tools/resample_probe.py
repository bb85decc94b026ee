#!/usr/bin/env python3
"""The 2x Mitchell resample of one 3840x2160 plane, N times (what the rocprofv3 passes for k_rs2d are collected from)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libsrcnn_amd as S
from libsrcnn_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S.init(0)
L = S.lib()
w, h = 3840, 2160
d_in = S.DeviceBuffer.from_numpy(synth.plane(h, w, synth.SEED0, "smooth"))
d_out = S.DeviceBuffer(4 * w * h * 4)
for _ in range(3):
    S.check(L.srcnn_resample_f32_dev(d_in.ptr, w, h, 2 * w, 2 * h, 2, d_out.ptr, None))
S.sync()
t0 = time.perf_counter()
for _ in range(n):
    S.check(L.srcnn_resample_f32_dev(d_in.ptr, w, h, 2 * w, 2 * h, 2, d_out.ptr, None))
S.sync()
print("resample 3840x2160 -> 7680x4320: %.4f ms per call (wall, back to back)" % ((time.perf_counter() - t0) * 1e3 / n))
