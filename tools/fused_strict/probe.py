#!/usr/bin/env python3
"""the strict FUSED kernel (tools/fused_strict/fused_strict.hip: layers 1+2+3 in one kernel, no layer-2 planes
in HBM), bit-for-bit against the golden planes and the production path, then timed against the production layer kernels on the
headline frame, alternating.  Run on the GPU box:  python3 tools/fused_strict/probe.py [--quick]"""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import libsrcnn_amd as S                      # noqa: E402
from libsrcnn_amd import synth                # noqa: E402

LIB = os.path.join(HERE, "libfused_strict.so")
SRC = os.path.join(HERE, "fused_strict.hip")


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        import shutil
        hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
                               "-shared", SRC, "-o", LIB])
    return C.CDLL(LIB)


def main():
    quick = "--quick" in sys.argv
    S.init(0)
    L = S.lib()
    F = build()
    F.fused_strict_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    w = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights_f32.bin"), dtype="<f4")
    assert w.size == 8129
    rc = F.fused_strict_init(w.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    print("device:", S.device_name(), "| fused kernel LDS %d bytes" % F.fused_strict_lds_bytes(), flush=True)

    def fused(y, chunks=0):
        """y: low-res plane -> Y' through the product's resampler + the fused kernel."""
        h, w_ = y.shape
        H, W = 2 * h, 2 * w_
        din = S.DeviceBuffer.from_numpy(y)
        dup = S.DeviceBuffer(H * W * 4)
        dout = S.DeviceBuffer(H * W * 4)
        S.check(L.srcnn_resample_f32_dev(din.ptr, w_, h, W, H, 2, dup.ptr, None))
        S.check(L.srcnn_memset_dev(dout.ptr, 0xFF, H * W * 4, None))
        S.sync()
        assert F.fused_strict_run(dup.ptr, W, H, dout.ptr, chunks, None) == 0
        S.sync()
        out = dout.to_numpy(np.float32, (H, W))
        for b in (din, dup, dout):
            b.free()
        return out

    def same(a, b):
        return bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))

    # ---- 1. the golden planes (made by the real reference) ----
    g = np.load(os.path.join(ROOT, "tests", "golden", "y_planes.npz"))
    names = sorted(k[:-3] for k in g.files if k.endswith("_in"))
    ok = True
    for n in names:
        got = fused(g[n + "_in"])
        e = same(got, g[n + "_out"])
        ok = ok and e
        print("golden %-28s %s -> %s  %s" % (n, g[n + "_in"].shape, got.shape, "bit-exact" if e else "MISMATCH max|d| %.3g" %
              float(np.nanmax(np.abs(got.astype(np.float64) - g[n + "_out"])))), flush=True)
    # ---- 2. shapes that straddle the strips (124 columns), the two-row steps, chunk seams, one-row / one-column planes ----
    rng = np.random.default_rng(5)
    shapes = [(1, 1), (1, 70), (70, 1), (2, 62), (3, 63), (31, 61), (32, 62), (33, 125), (64, 124), (65, 187), (127, 310), (260, 95)]
    shapes += [(int(rng.integers(1, 200)), int(rng.integers(1, 400))) for _ in range(6 if quick else 20)]
    bad = 0
    for i, (h, w_) in enumerate(shapes):
        y = synth.plane(h, w_, 1000 + i, "noise" if i & 1 else "smooth")
        want = S.y_upscale2x(y)
        for chunks in (0, 1, 3):
            if not same(fused(y, chunks), want):
                bad += 1
                print("MISMATCH shape", (h, w_), "chunks", chunks, flush=True)
    print("shape sweep: %d shapes x 3 chunkings, %d mismatches vs the production path" % (len(shapes), bad), flush=True)
    ok = ok and bad == 0

    # ---- 3. the headline frame: equality and time, alternating ----
    h, w_ = 2160, 3840
    H, W = 2 * h, 2 * w_
    y = synth.plane(h, w_, synth.SEED0, "smooth")
    din = S.DeviceBuffer.from_numpy(y)
    dup = S.DeviceBuffer(H * W * 4); dc2 = S.DeviceBuffer(32 * H * W * 4); do1 = S.DeviceBuffer(H * W * 4); do2 = S.DeviceBuffer(H * W * 4)
    S.check(L.srcnn_resample_f32_dev(din.ptr, w_, h, W, H, 2, dup.ptr, None))

    def two_kernels():
        S.check(L.srcnn_conv12_f32_dev(dup.ptr, W, H, dc2.ptr, None))
        S.check(L.srcnn_conv3_f32_dev(dc2.ptr, W, H, do1.ptr, None))

    def one_kernel(chunks=0):
        assert F.fused_strict_run(dup.ptr, W, H, do2.ptr, chunks, None) == 0

    def t(fn, n=4):
        fn(); S.sync()
        best = 1e9
        for _ in range(3):
            e0, e1 = S.Event(), S.Event(); e0.record()
            for _ in range(n):
                fn()
            e1.record(); best = min(best, e0.elapsed_ms(e1) / n)
        return best
    two_kernels(); one_kernel(); S.sync()
    a = do1.to_numpy(np.float32, (H, W)); b = do2.to_numpy(np.float32, (H, W))
    e = same(a, b)
    ok = ok and e
    print("3840x2160 -> 7680x4320: fused == production layers: %s  (sha %s)" % (e, hashlib.sha256(b.tobytes()).hexdigest()[:12]), flush=True)
    for rnd in range(2 if quick else 3):
        t2 = t(two_kernels)
        line = "round %d: production k_conv12_mfma + k_conv3 %.3f ms" % (rnd, t2)
        for chunks in (4, 8, 2):
            t1 = t(lambda: one_kernel(chunks))
            line += " | fused (%d chunks = %d workgroups) %.3f ms (%+.1f %%)" % (chunks, 62 * chunks, t1, (t1 / t2 - 1) * 100)
        print(line, flush=True)
    print("RESULT", "all bit-exact" if ok else "MISMATCHES")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
