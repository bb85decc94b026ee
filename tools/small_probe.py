#!/usr/bin/env python3
"""Small images: the Y path and ProcessSRCNN on planes / images far smaller than one round of the persistent layer-1+2 grid."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, libsrcnn_amd as S
from libsrcnn_amd import synth
S.init(0); L = S.lib()
for (h, w) in ((64, 64), (128, 128), (256, 256), (360, 640), (540, 960)):
    d_in = S.DeviceBuffer.from_numpy(synth.plane(h, w, 3, "smooth")); d_out = S.DeviceBuffer(4 * w * h * 4)
    for _ in range(20):
        S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_out.ptr, None))
    S.sync()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_out.ptr, None))
    S.sync()
    dt = (time.perf_counter() - t0) / n
    print("Y path %4dx%-4d -> x2: %.3f ms per frame (%.0f MPix/s)" % (w, h, dt * 1e3, 4 * w * h / 1e6 / dt), flush=True)
S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
fn = getattr(L, S.CXX_SYMBOLS[1])
for (h, w) in ((256, 256), (360, 640)):
    img = bench.synth_rgb(h, w, 5)
    ts = []
    for it in range(40):
        o, osz = C.c_void_p(), C.c_uint(0)
        t0 = time.perf_counter()
        assert fn(img.ctypes.data, w, h, 3, 2.0, C.byref(o), C.byref(osz), None, None) == 0
        ts.append(time.perf_counter() - t0)
        L.srcnn_delete_array(o)
    ts = sorted(ts[5:])
    print("ProcessSRCNN %4dx%-4d RGB x2: best %.3f med %.3f ms" % (w, h, ts[0] * 1e3, ts[len(ts) // 2] * 1e3), flush=True)
