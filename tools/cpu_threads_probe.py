#!/usr/bin/env python3
"""Which threads of the process burn host CPU while the library works?  Per-thread user+system time (from
/proc/self/task/*/stat) across (a) the PCIe-inclusive stream of page-locked 4K frames and (b) ProcessSRCNN on a 4K image.
    python tools/cpu_threads_probe.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, libsrcnn_amd as S

TICK = os.sysconf("SC_CLK_TCK")


def snap():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read()
        except OSError:
            continue
        name = f[f.index("(") + 1:f.rindex(")")]
        rest = f[f.rindex(")") + 2:].split()
        out[int(t)] = (name, (int(rest[11]), int(rest[12])))          # utime, stime
    return out


def report(what, a, b, wall, units):
    rows = []
    for t, (name, (u, s)) in b.items():
        u0, s0 = a.get(t, (name, (0, 0)))[1]
        if u - u0 + s - s0 > 0:
            rows.append(((u - u0) / TICK, (s - s0) / TICK, name, t, t not in a))
    rows.sort(reverse=True)
    tot = sum(r[0] + r[1] for r in rows)
    print("%s: wall %.1f ms per %s, CPU %.1f ms per %s over all threads" % (what, wall * 1e3 / units[0], units[1], tot * 1e3 / units[0], units[1]))
    for u, s, name, t, new in rows[:10]:
        print("    %-18s tid %-8d user %6.1f ms  sys %6.1f ms%s" % (name, t, u * 1e3 / units[0], s * 1e3 / units[0], "  (started during the run)" if new else ""))
    # threads that ended during the run are invisible here; the difference to process_time() shows them
    return tot


S.init(0); L = S.lib()
step, free = bench.host_stream_setup(S, 16)
step()
for rep in range(2):
    a = snap(); c0 = time.process_time(); t0 = time.perf_counter()
    for _ in range(4):
        step()
    wall = time.perf_counter() - t0; cpu = time.process_time() - c0; b = snap()
    tot = report("host stream (64 page-locked 4K frames)", a, b, wall, (64, "frame"))
    print("    process_time says %.1f ms per frame (threads that came and went: %.1f)" % (cpu * 1e3 / 64, (cpu - tot) * 1e3 / 64))
free()
S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
fn = getattr(L, S.CXX_SYMBOLS[1])
img = bench.synth_rgb(2160, 3840, 0x5C0DE000 + 2160)


def call():
    o, osz = C.c_void_p(), C.c_uint(0)
    assert fn(img.ctypes.data, 3840, 2160, 3, 2.0, C.byref(o), C.byref(osz), None, None) == 0
    L.srcnn_delete_array(o)


call(); call()
a = snap(); c0 = time.process_time(); t0 = time.perf_counter()
for _ in range(10):
    call()
wall = time.perf_counter() - t0; cpu = time.process_time() - c0; b = snap()
tot = report("ProcessSRCNN 4K RGB x2", a, b, wall, (10, "call"))
print("    process_time says %.1f ms per call (threads that came and went: %.1f)" % (cpu * 1e3 / 10, (cpu - tot) * 1e3 / 10))
