#!/usr/bin/env python3
"""Quick per-kernel timing on the GPU box (development aid, not the contract bench)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import libsrcnn_amd as S
from libsrcnn_amd import synth

def timeit(fn, n=5):
    fn(); S.sync()
    e0, e1 = S.Event(), S.Event()
    e0.record()
    for _ in range(n): fn()
    e1.record()
    return e0.elapsed_ms(e1) / n

def main():
    S.init(0)
    print(S.device_name())
    for (h, w) in [(1080, 1920), (2160, 3840)]:
        y = synth.plane(h, w, synth.SEED0, "smooth")
        din = S.DeviceBuffer.from_numpy(y)
        H, W = 2 * h, 2 * w
        dup = S.DeviceBuffer(H * W * 4)
        dc2 = S.DeviceBuffer(32 * H * W * 4)
        dout = S.DeviceBuffer(H * W * 4)
        L = S.lib()
        for mode in (S.MODE_STRICT, S.MODE_FAST):
            S.set_mode(mode)
            t_rs = timeit(lambda: S.check(L.srcnn_resample_f32_dev(din.ptr, w, h, W, H, 2, dup.ptr, None)))
            t_c12 = timeit(lambda: S.check(L.srcnn_conv12_f32_dev(dup.ptr, W, H, dc2.ptr, None)))
            t_c3 = timeit(lambda: S.check(L.srcnn_conv3_f32_dev(dc2.ptr, W, H, dout.ptr, None)))
            t_all = timeit(lambda: S.check(L.srcnn_y_upscale2x_f32_dev(din.ptr, w, h, dout.ptr, None)))
            mp = H * W / 1e6
            print("%dx%d->%dx%d mode=%s: resample %.3f ms  conv12 %.3f ms  conv3 %.3f ms  | whole %.3f ms = %.1f MPix/s  (%.2f TFLOP/s algorithmic)"
                  % (w, h, W, H, "strict" if mode == 0 else "fast", t_rs, t_c12, t_c3, t_all, mp / t_all * 1e3, 16064 * mp * 1e6 / (t_all * 1e-3) / 1e12))
        S.set_mode(S.MODE_STRICT)
        for b in (din, dup, dc2, dout): b.free()

if __name__ == "__main__":
    main()
