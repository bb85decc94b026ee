#!/usr/bin/env python3
"""How busy are the ROCm runtime's own threads while (a) nothing runs, (b) only kernels run (resident frames), (c) page-locked
frames stream through without / with hipGraph replay?  Per-thread CPU from /proc/self/task/*/stat, long-lived threads only."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, libsrcnn_amd as S
from libsrcnn_amd import synth
TICK = os.sysconf("SC_CLK_TCK")


def snap():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read()
        except OSError:
            continue
        rest = f[f.rindex(")") + 2:].split()
        out[int(t)] = (int(rest[11]), int(rest[12]))
    return out


def measure(what, fn):
    a = snap(); t0 = time.perf_counter()
    fn()
    wall = time.perf_counter() - t0; b = snap()
    me = os.getpid()
    rows = sorted(((b[t][0] - a[t][0] + b[t][1] - a[t][1], b[t][0] - a[t][0], b[t][1] - a[t][1], t) for t in b if t in a), reverse=True)[:3]
    print("%-52s wall %7.1f ms | busiest long-lived threads: %s" % (what, wall * 1e3, ", ".join(
        "%s %.0f%% (user %.0f sys %.0f ms)" % ("main" if t == me else "tid %d" % t, 100 * tot / TICK / wall, u / TICK * 1e3, s / TICK * 1e3)
        for tot, u, s, t in rows)), flush=True)


S.init(0); L = S.lib()
measure("idle (sleep 1 s)", lambda: time.sleep(1.0))
w, h, F = 3840, 2160, 4
d_in = S.DeviceBuffer(F * w * h * 4); d_out = S.DeviceBuffer(F * 4 * w * h * 4)
for f in range(F):
    d_in.upload(synth.plane(h, w, 7 + f, "smooth"), offset=f * w * h * 4)
def resident():
    for _ in range(25):
        S.check(L.srcnn_y_upscale2x_f32_batch_dev(d_in.ptr, w, h, F, d_out.ptr, None))
    while L.srcnn_stream_query(None) if hasattr(L, "srcnn_stream_query") else False:
        time.sleep(0.001)
    t_end = time.perf_counter() + 1.2            # ~1 s of kernels: sleep instead of a spinning sync so that only the runtime shows
    time.sleep(max(0.0, t_end - time.perf_counter()))
    S.sync()
resident()
measure("resident frames: 100 frames of kernels, host asleep", resident)
pin_in = L.srcnn_host_alloc_pinned(16 * w * h * 4); pin_out = L.srcnn_host_alloc_pinned(16 * 4 * w * h * 4)
for g in (0, 1):
    S.check(L.srcnn_y_upscale2x_f32_stream(pin_in, w, h, 16, pin_out, g))
    S.check(L.srcnn_y_upscale2x_f32_stream(pin_in, w, h, 16, pin_out, g))
    measure("page-locked frame stream, 64 frames, graph=%d" % g,
            lambda: [S.check(L.srcnn_y_upscale2x_f32_stream(pin_in, w, h, 16, pin_out, g)) for _ in range(4)])
L.srcnn_host_free_pinned(pin_in); L.srcnn_host_free_pinned(pin_out)
import numpy as np
S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
img = bench.synth_rgb(2160, 3840, 0x5C0DE000 + 2160)
res = np.empty((4320, 7680, 3), np.uint8)
fn = getattr(L, S.CXX_SYMBOLS[1])
def reused():
    for _ in range(20):
        S.check(L.srcnn_process_u8(img.ctypes.data, 3840, 2160, 3, 2.0, 2, res.ctypes.data, None))
def fresh():
    for _ in range(20):
        o, osz = C.c_void_p(), C.c_uint(0)
        assert fn(img.ctypes.data, 3840, 2160, 3, 2.0, C.byref(o), C.byref(osz), None, None) == 0
        L.srcnn_delete_array(o)
reused(); fresh()
c0 = time.process_time(); measure("srcnn_process_u8 4K RGB x2, 20 calls, reused result", reused); print("    process CPU per call: %.1f ms" % ((time.process_time() - c0) * 50))
c0 = time.process_time(); measure("ProcessSRCNN 4K RGB x2, 20 calls, fresh result", fresh); print("    process CPU per call: %.1f ms" % ((time.process_time() - c0) * 50))
