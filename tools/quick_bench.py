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
        for mode in (S.MODE_STRICT, S.MODE_FAST, S.MODE_FAST_F16):
            S.set_mode(mode)
            t_rs = timeit(lambda: S.check(L.srcnn_resample_f32_dev(din.ptr, w, h, W, H, 2, dup.ptr, None)))
            t_c12 = timeit(lambda: S.check(L.srcnn_conv12_f32_dev(dup.ptr, W, H, dc2.ptr, None)))
            t_c3 = timeit(lambda: S.check(L.srcnn_conv3_f32_dev(dc2.ptr, W, H, dout.ptr, None)))
            t_all = timeit(lambda: S.check(L.srcnn_y_upscale2x_f32_dev(din.ptr, w, h, dout.ptr, None)))
            mp = H * W / 1e6
            print("%dx%d->%dx%d mode=%s: resample %.3f ms  conv12 %.3f ms  conv3 %.3f ms  | whole %.3f ms = %.1f MPix/s  (%.2f TFLOP/s algorithmic)"
                  % (w, h, W, H, ("strict", "fast", "fast_f16")[mode], t_rs, t_c12, t_c3, t_all, mp / t_all * 1e3, 16064 * mp * 1e6 / (t_all * 1e-3) / 1e12))
        S.set_mode(S.MODE_STRICT)
        for b in (din, dup, dc2, dout): b.free()

if __name__ == "__main__" and "--dropin" not in sys.argv:
    main()


def dropin_latency():
    """Wall time of the ProcessSRCNN drop-in call (host u8 in, new[] u8 out) on a 1080p RGB image."""
    import time
    S.init(0)
    rng = np.random.default_rng(0)
    for (h, w) in ((256, 256), (1080, 1920), (2160, 3840)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
        S.ProcessSRCNN(img, w, h, 3, 2.0)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            rc, out, conv = S.ProcessSRCNN(img, w, h, 3, 2.0)
            ts.append(time.perf_counter() - t0)
        print("ProcessSRCNN %dx%dx3 x2 (host u8 -> host u8, incl. PCIe, new[] and the ctypes copy): best %.2f ms = %.0f MPix/s"
              % (w, h, min(ts) * 1e3, 4 * w * h / 1e6 / min(ts)))


if __name__ == "__main__" and "--dropin" in sys.argv:
    dropin_latency()
