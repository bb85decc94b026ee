#!/usr/bin/env python3
"""Does keeping the GPU trivially busy between calls keep its clocks up?  srcnn_process_u8 (4K RGB x2, reused buffer) with a
10 ms host pause between calls, (a) GPU idle during the pause, (b) a background thread launching tiny resamples back to back
on another stream during the pause.  Per-call stage device time from the library's timers."""
import ctypes as C, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, libsrcnn_amd as S
from libsrcnn_amd import synth
S.init(0); L = S.lib()
img = bench.synth_rgb(2160, 3840, 0x5C0DE000 + 2160)
S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
out = np.empty((4320, 7680, 3), np.uint8)
st = C.c_void_p(); S.check(L.srcnn_stream_create(C.byref(st)))
tiny_in = S.DeviceBuffer.from_numpy(synth.plane(64, 64, 1, "smooth")); tiny_out = S.DeviceBuffer(128 * 128 * 4)
busy = threading.Event(); stop = False


def warmer():
    while not stop:
        if busy.is_set():
            for _ in range(20):
                L.srcnn_resample_f32_dev(tiny_in.ptr, 64, 64, 128, 128, 2, tiny_out.ptr, st)
            L.srcnn_stream_sync(st)
        else:
            time.sleep(0.0002)


th = threading.Thread(target=warmer, daemon=True); th.start()


def call():
    S.check(L.srcnn_process_u8(img.ctypes.data, 3840, 2160, 3, 2.0, 2, out.ctypes.data, None))


for pause, warm in ((0.0, False), (0.010, False), (0.010, True), (0.050, False), (0.050, True)):
    call(); call()
    S.profile_reset(); S.profile_enable(True)
    ts = []
    for _ in range(10):
        busy.clear()
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
        if pause:
            if warm: busy.set()
            time.sleep(pause)
    busy.clear()
    S.profile_enable(False)
    p = S.profile_read()
    print("pause %.3f s, %-22s wall best %.2f med %.2f ms | device per call: %s" % (
        pause, "GPU kept busy" if warm else "GPU idle in the pause", min(ts) * 1e3, sorted(ts)[5] * 1e3,
        {k: round(v[0] / 10, 3) for k, v in p.items() if k in ("conv12", "conv3")}), flush=True)
stop = True
