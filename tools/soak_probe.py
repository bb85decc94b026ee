#!/usr/bin/env python3
"""Soak: N ProcessSRCNN calls (4K RGB x2, fresh result each) then N/2 frame-stream steps; prints wall-time percentiles and
the process's RSS / thread count / open fds before and after (leaks show up as growth).   python tools/soak_probe.py [N]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, libsrcnn_amd as S


def vitals():
    st = open("/proc/self/status").read()
    rss = int([l for l in st.splitlines() if l.startswith("VmRSS")][0].split()[1]) / 1024
    thr = int([l for l in st.splitlines() if l.startswith("Threads")][0].split()[1])
    return "RSS %.0f MB, %d threads, %d fds" % (rss, thr, len(os.listdir("/proc/self/fd")))


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S.init(0); L = S.lib()
S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
fn = getattr(L, S.CXX_SYMBOLS[1])
img = bench.synth_rgb(2160, 3840, 0x5C0DE000 + 2160)


def call():
    o, osz = C.c_void_p(), C.c_uint(0)
    t0 = time.perf_counter()
    rc = fn(img.ctypes.data, 3840, 2160, 3, 2.0, C.byref(o), C.byref(osz), None, None)
    dt = time.perf_counter() - t0
    assert rc == 0 and osz.value == 4 * 2160 * 3840 * 3
    L.srcnn_delete_array(o)
    return dt


for _ in range(5):
    call()
print("before:", vitals(), flush=True)
ts = []
for i in range(n):
    ts.append(call())
    if i % 250 == 249:
        print("  ... %d calls, %s" % (i + 1, vitals()), flush=True)
ts.sort()
print("%d x ProcessSRCNN 4K RGB: min %.2f med %.2f p90 %.2f p99 %.2f max %.2f ms" % (
    n, ts[0] * 1e3, ts[n // 2] * 1e3, ts[int(n * 0.9)] * 1e3, ts[int(n * 0.99)] * 1e3, ts[-1] * 1e3))
step, free = bench.host_stream_setup(S, 16, 0)
step()
t0 = time.perf_counter()
m = max(1, n // 40)
for i in range(m):
    step()
dt = time.perf_counter() - t0
free()
print("%d frames through the page-locked stream (plain launches): %.2f ms per frame" % (16 * m, dt * 1e3 / (16 * m)))
print("after: ", vitals())
