#!/usr/bin/env python3
"""is the +15-18 % the layer kernels cost inside a ProcessSRCNN call CLOCK, or something else?

Every k_conv12_mfma launch stamps the shader-cycle counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) over
the lifetime of its first workgroup (srcnn_debug_clock_probe): cycles / ticks x 100 = MHz of that launch, ticks / 100 = its
duration in microseconds.  A launch's WORK is its rounds of the persistent grid (tiles / 512 resident workgroups), so
"cycles per round" is the clock-independent cost: if it is the same inside ProcessSRCNN and in the resident batch, the whole
difference is MHz; if it is higher, something else (cold weights, partly filled rounds, contention with the copies) is.

Compared: (a) resident 4K frames back to back (the bench workload); (b) the same frame as the seven bands ProcessSRCNN cuts,
back to back on one stream; (c) ProcessSRCNN 4K RGB calls back to back (fresh result each, like the drop-in); (d) the same into
a reused buffer (srcnn_process_u8); (e) two asynchronous jobs in flight (srcnn_process_u8_begin/_wait).

    python3 tools/process_clock_probe.py  > profiles/r04_process_clock.txt"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import libsrcnn_amd as S
from libsrcnn_amd import synth

S.init(0)
L = S.lib()
W, H = 3840, 2160
DW, DH = 2 * W, 2 * H
GRID = 512                                      # resident workgroups of the layer-1+2 kernel on 256 CUs


def rounds(rows):
    """rounds of the persistent grid for a band of `rows` output rows (+2 halo rows of layer 2 per interior side ~ +4)"""
    return -(-rows // 16) * (DW // 64) / GRID


def summarize(name, recs, rows_of=None, wall_ms=None):
    """recs: [(MHz, us)] per launch"""
    mhz = np.array([r[0] for r in recs]); us = np.array([r[1] for r in recs])
    line = "%-46s launches %4d  MHz: median %6.0f  min %6.0f  max %6.0f | launch us: median %8.1f  sum/call %9.1f" % (
        name, len(recs), np.median(mhz), mhz.min(), mhz.max(), np.median(us), us.sum() / max(1, rows_of or 1))
    if wall_ms is not None:
        line += " | wall %.2f ms" % wall_ms
    print(line, flush=True)
    return mhz, us


ASYNC_ONLY = "--async-only" in sys.argv
print("# device:", S.device_name(), "| SRCNN_ASYNC_CHAIN =", os.environ.get("SRCNN_ASYNC_CHAIN", "(default: chained)"))
y = synth.plane(H, W, synth.SEED0, "smooth")
d_in = S.DeviceBuffer.from_numpy(y)
d_out = S.DeviceBuffer(4 * W * H * 4)

if not ASYNC_ONLY:
    # ---- (a) resident frames back to back ----
    for _ in range(4):
        S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, W, H, d_out.ptr, None))
    S.sync()
    S.clock_probe(True)
    N = 24
    for _ in range(N):
        S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, W, H, d_out.ptr, None))
    S.sync()
    ra = S.clock_read()
    S.clock_probe(False)
    mhz_a, us_a = summarize("(a) resident 4K frames, back to back", ra, N)
    cyc_round_a = np.median(mhz_a * us_a) / rounds(DH)
    print("    cycles per round of the grid: %.0f   (frame = %.2f rounds)" % (cyc_round_a, rounds(DH)))

    # ---- (b) the same frame as ProcessSRCNN's bands, back to back on one stream ----
    cuts = (C.c_uint * 16)()
    nb = L.srcnn_debug_band_plan(0, DH, DW, 0, cuts, 16)
    cuts = [cuts[i] for i in range(nb)]
    print("# band plan of a 7680x4320 output:", cuts)
    for _ in range(2):
        for a, b in zip(cuts[:-1], cuts[1:]):
            S.check(L.srcnn_y_upscale2x_f32_band_dev(d_in.ptr, W, H, a, b - a, d_out.ptr + a * DW * 4, None))
    S.sync()
    S.clock_probe(True)
    for _ in range(N):
        for a, b in zip(cuts[:-1], cuts[1:]):
            S.check(L.srcnn_y_upscale2x_f32_band_dev(d_in.ptr, W, H, a, b - a, d_out.ptr + a * DW * 4, None))
    S.sync()
    rb = S.clock_read()
    S.clock_probe(False)
    summarize("(b) the frame as %d bands, back to back" % (nb - 1), rb, N)
    per_band = len(cuts) - 1
    for k in range(per_band):
        sel = rb[k::per_band]
        m = np.array([r[0] for r in sel]); u = np.array([r[1] for r in sel])
        rows = cuts[k + 1] - cuts[k] + (2 if k else 0) + (2 if k + 1 < per_band else 0)
        print("    band %d rows %4d: MHz median %6.0f  us median %8.1f  cycles/round %.0f" % (k, cuts[k + 1] - cuts[k], np.median(m), np.median(u),
                                                                                           np.median(m * u) / rounds(rows)))

# ---- (c)/(d)/(e) ProcessSRCNN ----
img = bench.synth_rgb(H, W, 0x5C0DE000 + 2160)
S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
fn = getattr(L, S.CXX_SYMBOLS[1])


def call_dropin():
    o, osz = C.c_void_p(), C.c_uint(0)
    rc = fn(img.ctypes.data, W, H, 3, 2.0, C.byref(o), C.byref(osz), None, None)
    assert rc == 0
    L.srcnn_delete_array(o)


out = np.empty((DH, DW, 3), np.uint8)
out2 = np.empty((DH, DW, 3), np.uint8)


def call_reused():
    S.check(L.srcnn_process_u8(img.ctypes.data, W, H, 3, 2.0, 2, out.ctypes.data, None))


def run_calls(name, call, reps=12):
    for _ in range(3):
        call()
    S.clock_probe(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    wall = (time.perf_counter() - t0) * 1e3 / reps
    recs = S.clock_read()
    S.clock_probe(False)
    mhz, us = summarize(name, recs, reps, wall)
    nb_call = len(recs) // reps
    for k in range(nb_call):
        sel = recs[k::nb_call]
        m = np.array([r[0] for r in sel]); u = np.array([r[1] for r in sel])
        print("    band %d: MHz median %6.0f (min %6.0f)  us median %8.1f" % (k, np.median(m), m.min(), np.median(u)))
    return wall


if not ASYNC_ONLY:
    run_calls("(c) ProcessSRCNN 4K RGB, fresh result each call", call_dropin)
run_calls("(d) srcnn_process_u8 into a reused buffer", call_reused)

# (f) page-locked caller buffers: no staging memcpy, no fan-out
pin_img = S.PinnedArray(img.shape); pin_img.array[...] = img
pin_a = S.PinnedArray(out.shape); pin_b = S.PinnedArray(out.shape)


def call_pinned():
    S.check(L.srcnn_process_u8(pin_img.array.ctypes.data, W, H, 3, 2.0, 2, pin_a.array.ctypes.data, None))


run_calls("(f) srcnn_process_u8, page-locked source and result", call_pinned)
assert np.array_equal(pin_a.array, out)


def run_async(name, src, bufs, reps=24, depth=2):
    for _ in range(2):
        call_reused()
    S.clock_probe(True)
    t0 = time.perf_counter()
    jobs = []
    for i in range(reps):
        j = C.c_void_p()
        S.check(L.srcnn_process_u8_begin(src.ctypes.data, W, H, 3, 2.0, 2, bufs[i % len(bufs)].ctypes.data, None, C.byref(j)))
        jobs.append(j)
        if len(jobs) == depth:
            S.check(L.srcnn_process_u8_wait(jobs.pop(0)))
    while jobs:
        S.check(L.srcnn_process_u8_wait(jobs.pop(0)))
    wall = (time.perf_counter() - t0) * 1e3 / reps
    recs = S.clock_read()
    S.clock_probe(False)
    summarize(name, recs, reps, wall)
    print("    images per second: %.1f  = %.2f GPix/s through host u8 buffers" % (1e3 / wall, DW * DH / wall / 1e6))


run_async("(g) two asynchronous jobs in flight, page-locked buffers", pin_img.array, [pin_a.array, pin_b.array])
assert np.array_equal(pin_a.array, out) and np.array_equal(pin_b.array, out)

# (e) two jobs in flight
bufs = [out, out2]
for _ in range(2):
    call_reused()
S.clock_probe(True)
reps = 24
t0 = time.perf_counter()
jobs = []
for i in range(reps):
    j = C.c_void_p()
    S.check(L.srcnn_process_u8_begin(img.ctypes.data, W, H, 3, 2.0, 2, bufs[i & 1].ctypes.data, None, C.byref(j)))
    jobs.append(j)
    if len(jobs) == 2:
        S.check(L.srcnn_process_u8_wait(jobs.pop(0)))
while jobs:
    S.check(L.srcnn_process_u8_wait(jobs.pop(0)))
wall = (time.perf_counter() - t0) * 1e3 / reps
re_ = S.clock_read()
S.clock_probe(False)
summarize("(e) two asynchronous jobs in flight (begin/wait)", re_, reps, wall)
print("    images per second: %.1f  = %.2f GPix/s through host u8 buffers" % (1e3 / wall, DW * DH / wall / 1e6))
assert np.array_equal(out, out2)
