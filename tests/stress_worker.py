"""Worker for tests/test_gpu_stress.py and tools/dma_stress.sh: the launch pattern that once ended in a silent SIGABRT
(DESIGN history, round 3: small-shape parity launches, each followed at once by the NEXT case's host-to-device copy) repeated
thousands of times over random shapes 1..130, every result compared bit for bit with the oracle.  Runs in its OWN process so
that the kernel selection comes from the environment (SRCNN_CONV12_DMA, SRCNN_CONV3_WDMA, SRCNN_CONV12_SPREAD ...) and
so that a runtime abort leaves its stderr in a file instead of in pytest's capture buffer.  Test infrastructure: uses oracle/.

    python tests/stress_worker.py ITERATIONS SEED [POOL [POOL_FILE]]      -> one JSON line; exit code 1 on any mismatch"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import libsrcnn_amd as S          # noqa: E402
from libsrcnn_amd import synth     # noqa: E402
import oracle                      # noqa: E402


def make_pool(pool_n, seed):
    """pool_n float cases (plane, the oracle's layer-2 planes, the oracle's result) and pool_n // 8 small images with the
    oracle's ProcessSRCNN bytes."""
    rng = np.random.default_rng([seed, 77])
    orc = oracle.Oracle()
    pool = []
    for k in range(pool_n):
        h, w = int(rng.integers(1, 131)), int(rng.integers(1, 131))
        if k % 9 == 8:                                  # a few cases with several column tiles / a ragged last tile row
            h, w = int(rng.integers(17, 80)), int(rng.integers(130, 400))
        y = synth.plane(h, w, int(rng.integers(0, 1 << 30)), "noise" if k % 2 else "smooth")
        if k % 11 == 10:
            y *= np.float32(rng.choice([0.0, 1e-3, 4.0, -1.0]))
        out, _, _, c2 = orc.y_path(y, taps=True)
        pool.append({"y": y, "c2": np.ascontiguousarray(c2), "out": out})
    imgs = []
    for k in range(max(4, pool_n // 8)):
        h, w, d = int(rng.integers(1, 100)), int(rng.integers(1, 130)), int(rng.choice([3, 4]))
        img = rng.integers(0, 256, (h, w, d), dtype=np.uint8)
        imgs.append((img,) + tuple(orc.process(img, 2.0)))
    return pool, imgs


def save_pool(path, pool_n, seed):
    pool, imgs = make_pool(pool_n, seed)
    d = {"n": len(pool), "ni": len(imgs)}
    for k, p in enumerate(pool):
        d["y%d" % k] = p["y"]; d["c2_%d" % k] = p["c2"]; d["out%d" % k] = p["out"]
    for k, (img, rgb, conv) in enumerate(imgs):
        d["img%d" % k] = img; d["rgb%d" % k] = rgb; d["conv%d" % k] = conv
    np.savez(path, **d)


def main(iters, seed, pool_n, pool_file=None):
    rng = np.random.default_rng(seed)
    S.init(0)
    L = S.lib()
    t0 = time.time()
    # ---- a pool of cases with their oracle answers (the oracle is the slow part: the parent test computes the pool once,
    #      with make_pool() below, and hands the file to every child) ----
    if pool_file and os.path.exists(pool_file):
        z = np.load(pool_file)
        n = int(z["n"]); ni = int(z["ni"])
        pool = [{"y": z["y%d" % k], "c2": z["c2_%d" % k], "out": z["out%d" % k]} for k in range(n)]
        imgs = [(z["img%d" % k], z["rgb%d" % k], z["conv%d" % k]) for k in range(ni)]
    else:
        pool, imgs = make_pool(pool_n, seed)
    t_pool = time.time() - t0

    # ---- device buffers sized for the largest case, reused (no allocation inside the loop) ----
    max_in = max(p["y"].size for p in pool)
    d_in = [S.DeviceBuffer(max_in * 4) for _ in range(2)]
    d_out = S.DeviceBuffer(max_in * 16)
    d_c2 = S.DeviceBuffer(max_in * 4 * 32 * 4)
    junk = rng.integers(0, 255, 1 << 20, dtype=np.uint8)             # pageable host memory for the interleaved copies
    d_junk = S.DeviceBuffer(junk.nbytes)
    bad = []
    counts = {"y2x_host": 0, "y2x_dev": 0, "band": 0, "conv3": 0, "process": 0}
    t0 = time.time()
    for it in range(iters):
        p = pool[int(rng.integers(0, len(pool)))]
        y = p["y"]
        h, w = y.shape
        kind = int(rng.integers(0, 10))
        what = ""
        if kind < 3:                                    # host-pointer call: H2D, kernels, D2H inside the library
            got = S.y_upscale2x(y); want = p["out"]; what = "y2x_host"
        elif kind < 6:
            # device call, and the NEXT case's H2D is issued while these kernels are still in flight (null stream: the copy
            # is ordered behind them) -- the exact place the round-3 abort surfaced
            what = "y2x_dev"
            d_in[it & 1].upload(y)
            S.check(L.srcnn_y_upscale2x_f32_dev(d_in[it & 1].ptr, w, h, d_out.ptr, None))
            nxt = pool[int(rng.integers(0, len(pool)))]["y"]
            d_in[(it + 1) & 1].upload(nxt)
            n = int(rng.integers(1, junk.size))
            S.check(L.srcnn_memcpy_h2d(d_junk.ptr, junk.ctypes.data, n, None))
            got = d_out.to_numpy(np.float32, (2 * h, 2 * w)); want = p["out"]
        elif kind < 8:                                  # a band of the frame
            what = "band"
            r0 = int(rng.integers(0, 2 * h)); rows = int(rng.integers(1, 2 * h - r0 + 1))
            d_in[0].upload(y)
            S.check(L.srcnn_y_upscale2x_f32_band_dev(d_in[0].ptr, w, h, r0, rows, d_out.ptr, None))
            S.check(L.srcnn_memcpy_h2d(d_junk.ptr, junk.ctypes.data, int(rng.integers(1, 4096)), None))
            got = d_out.to_numpy(np.float32, (rows, 2 * w)); want = p["out"][r0:r0 + rows]
        elif kind < 9:                                  # layer 3 alone on the oracle's layer-2 planes (its weight staging)
            what = "conv3"
            d_c2.upload(p["c2"])
            S.check(L.srcnn_conv3_f32_dev(d_c2.ptr, 2 * w, 2 * h, d_out.ptr, None))
            S.check(L.srcnn_memcpy_h2d(d_junk.ptr, junk.ctypes.data, int(rng.integers(1, 4096)), None))
            got = d_out.to_numpy(np.float32, (2 * h, 2 * w)); want = p["out"]
        else:                                           # the drop-in surface on a small image
            what = "process"
            img, want_rgb, want_conv = imgs[int(rng.integers(0, len(imgs)))]
            got_rgb, got_conv = S.process_u8(img, 2.0)
            counts[what] += 1
            if not (np.array_equal(got_rgb, want_rgb) and np.array_equal(got_conv, want_conv)):
                bad.append("it %d process %s" % (it, img.shape))
            continue
        counts[what] += 1
        if got.shape != want.shape or not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
            bad.append("it %d %s %dx%d" % (it, what, w, h))
        if it % 500 == 499:
            print("  ... %d iterations, %d mismatches" % (it + 1, len(bad)), file=sys.stderr, flush=True)
    S.sync()
    res = {"iterations": iters, "pool": pool_n, "mismatches": len(bad), "first_bad": bad[:5], "counts": counts,
           "pool_s": round(t_pool, 1), "loop_s": round(time.time() - t0, 1),
           "env": {k: v for k, v in os.environ.items() if k.startswith("SRCNN_")}}
    print(json.dumps(res), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                  int(sys.argv[3]) if len(sys.argv) > 3 else 48, sys.argv[4] if len(sys.argv) > 4 else None))
