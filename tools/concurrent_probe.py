#!/usr/bin/env python3
"""T host threads upscaling 4K RGB images concurrently: through ProcessSRCNN (fresh new[] result per call) and through
srcnn_process_u8 (caller-owned, reused result buffers).  Prints ms per image and the device time of the layer kernels.
    python tools/concurrent_probe.py [threads ...]"""
import ctypes as C, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, libsrcnn_amd as S
S.init(0); L = S.lib()
S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
fn = getattr(L, S.CXX_SYMBOLS[1])
h, w, per = 2160, 3840, 6
counts = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]


def run(T, reuse):
    imgs = [bench.synth_rgb(h, w, 0x5C0DE100 + t) for t in range(T)]
    outs = [np.empty((2 * h, 2 * w, 3), np.uint8) for _ in range(T)] if reuse else None

    def worker(t, n):
        for _ in range(n):
            if reuse:
                S.check(L.srcnn_process_u8(imgs[t].ctypes.data, w, h, 3, 2.0, 2, outs[t].ctypes.data, None))
            else:
                o, osz = C.c_void_p(), C.c_uint(0)
                assert fn(imgs[t].ctypes.data, w, h, 3, 2.0, C.byref(o), C.byref(osz), None, None) == 0
                L.srcnn_delete_array(o)

    def go(n):
        th = [threading.Thread(target=worker, args=(t, n)) for t in range(T)]
        t0 = time.perf_counter()
        [x.start() for x in th]; [x.join() for x in th]
        return time.perf_counter() - t0
    go(2)
    S.profile_reset(); S.profile_enable(True)
    c0 = time.process_time()
    dt = go(per)
    cpu = time.process_time() - c0
    S.profile_enable(False)
    p = S.profile_read()
    n = T * per
    print("%-16s threads %d: %.2f ms per image (%.0f MPix/s), host cpu %.1f ms per image, device per image: %s" % (
        "process_u8" if reuse else "ProcessSRCNN", T, dt * 1e3 / n, n * 4 * h * w / 1e6 / dt, cpu * 1e3 / n,
        {k: round(v[0] / n, 3) for k, v in p.items()}), flush=True)


for T in counts:
    run(T, True)
    run(T, False)
