#!/usr/bin/env python3
"""what does a memory-constrained caller pay?  The strict path keeps the 32 layer-2 planes of a band in
scratch (128 B per output pixel: 4.25 GB for a whole 3840x2160 -> 7680x4320 frame); srcnn_set_workspace_limit caps that
and the frame is produced in row bands, bit-identically (halo rows recomputed per band, short bands fill the persistent
layer-1+2 grid badly).  The alternative is the fused strict prototype (tools/fused_strict/: no layer-2 planes at all,
11.6 ms per frame, +22 %).  This prints ms per 8K frame against the cap for the banded product path and the fused
prototype's time beside it: the crossover is where the prototype would start to pay.

    python3 tools/lowmem_probe.py > profiles/r06_lowmem.txt       (on the GPU box)
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libsrcnn_amd as S                      # noqa: E402
from libsrcnn_amd import synth                # noqa: E402


def main():
    S.init(0)
    L = S.lib()
    h, w = 2160, 3840
    H, W = 2 * h, 2 * w
    din = S.DeviceBuffer.from_numpy(synth.plane(h, w, synth.SEED0, "smooth"))
    dout = S.DeviceBuffer(H * W * 4)

    def timed(fn, reps=3, n=3):
        fn(); S.sync()
        best = 1e9
        for _ in range(reps):
            e0, e1 = S.Event(), S.Event()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            best = min(best, e0.elapsed_ms(e1) / n)
        return best

    print("# tools/lowmem_probe.py on %s: one 3840x2160 -> 7680x4320 frame, strict, resident; ms per frame (HIP events, best of 3 x 3)" % S.device_name().strip())
    print("# layer-2 scratch of the whole frame: %.2f GB (128 B per output pixel); a band's scratch = 128 B x 7680 x (rows + 4)" % (128 * H * W / 1e9))
    prev = L.srcnn_set_workspace_limit(16 << 30)
    ref_sha = None
    rows = []
    try:
        for cap_mb in (16384, 4096, 2048, 1024, 512, 256, 128, 64, 32, 16):
            L.srcnn_set_workspace_limit(cap_mb << 20)
            S.check(L.srcnn_trim())                                   # scratch really is what the cap allows
            ms = timed(lambda: S.check(L.srcnn_y_upscale2x_f32_dev(din.ptr, w, h, dout.ptr, None)))
            sha = hashlib.sha256(dout.to_numpy(np.float32, (H, W)).tobytes()).hexdigest()[:12]
            ref_sha = ref_sha or sha
            band_rows = min(H, max(16, (cap_mb << 20) // (128 * W) - 4))
            rows.append((cap_mb, band_rows, ms, sha == ref_sha))
            print("cap %6d MiB  band <= %5d rows (~%3d bands)  %7.3f ms  = %5.0f MPix/s  %+6.1f %%  %s"
                  % (cap_mb, band_rows, -(-H // band_rows), ms, H * W / 1e3 / ms, (ms / rows[0][2] - 1) * 100, "bit-identical" if sha == ref_sha else "DIFFERS"), flush=True)
    finally:
        L.srcnn_set_workspace_limit(prev)
    # the fused prototype on the same frame (its own resample-free entry: an upscaled plane in)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools", "fused_strict"))
        import probe as fp
        F = fp.build()
        F.fused_strict_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        wts = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights_f32.bin"), dtype="<f4")
        assert F.fused_strict_init(wts.ctypes.data_as(C.c_void_p)) == 0
        dup = S.DeviceBuffer(H * W * 4)
        S.check(L.srcnn_resample_f32_dev(din.ptr, w, h, W, H, 2, dup.ptr, None))
        rs = timed(lambda: S.check(L.srcnn_resample_f32_dev(din.ptr, w, h, W, H, 2, dup.ptr, None)))
        fu = timed(lambda: F.fused_strict_run(dup.ptr, W, H, dout.ptr, 0, None))
        sha = hashlib.sha256(dout.to_numpy(np.float32, (H, W)).tobytes()).hexdigest()[:12]
        total = rs + fu
        print("fused strict prototype (0 B of layer-2 scratch): resample %.3f + fused kernel %.3f = %7.3f ms = %5.0f MPix/s  %+6.1f %%  %s"
              % (rs, fu, total, H * W / 1e3 / total, (total / rows[0][2] - 1) * 100, "bit-identical" if sha == ref_sha else "DIFFERS"))
        cross = [r for r in rows if r[2] > total]
        print("# crossover: banding is faster than the fused prototype down to a cap of %s"
              % ("%d MiB (first slower: %d MiB)" % (min(r[0] for r in rows if r[2] <= total), max(r[0] for r in cross)) if cross else "16 MiB (never slower in this table)"))
    except Exception as e:                                            # noqa: BLE001
        print("# fused prototype not measured: %r" % (e,))


if __name__ == "__main__":
    main()
