"""Worker for tests/test_gpu_node.py: runs in its OWN process because contexts are process-wide (the session fixture of
the other GPU tests binds one context on device 0).  Prints one JSON object.  Test infrastructure: may use oracle/."""
import ctypes as C
import hashlib
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import libsrcnn_amd as S          # noqa: E402
from libsrcnn_amd import synth     # noqa: E402
import oracle                      # noqa: E402


def image(h, w, d, seed):
    rng = np.random.default_rng(seed)
    base = synth.plane(h, w, synth.SEED0 + seed, "smooth")
    img = np.empty((h, w, d), np.uint8)
    for k in range(d):
        img[..., k] = np.clip(base * (0.55 + 0.15 * k) + rng.integers(0, 40, base.shape), 0, 255).astype(np.uint8)
    return img


def devices(nctx):
    """Context k -> HIP device k mod (visible devices): K virtual contexts on a one-GPU box, real peers on a multi-GPU node."""
    nd = max(1, S.device_count())
    return [k % nd for k in range(nctx)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def lanes():
    t, l = C.c_int(), C.c_int()
    S.lib().srcnn_debug_counts(C.byref(t), C.byref(l))
    return l.value


def cmd_process(nctx):
    """ProcessSRCNN dealt over `nctx` virtual contexts on device 0: vs the oracle (medium images, several filters / ratios /
    depths), vs the single-context result (4K), and from 4 host threads at once."""
    res = {}
    orc = oracle.Oracle()
    n = S.init_devices(devices(nctx))
    res["contexts"] = n
    S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
    cases = [(image(1100, 1200, 3, 11), 2.0, 2), (image(900, 1400, 4, 12), 1.5, 3), (image(1300, 1000, 3, 13), 3.0, 1),
             (image(1201, 1111, 4, 14), 2.0, 2)]
    ok = []
    for img, m, filt in cases:
        want_rgb, want_conv = orc.process(img, m, filt)
        got_rgb, got_conv = S.process_u8(img, m, filt)
        ok.append(bool(np.array_equal(got_rgb, want_rgb) and np.array_equal(got_conv, want_conv)))
    res["vs_oracle"] = ok
    res["lanes_after_oracle_cases"] = lanes()
    # 4 host threads, each a ProcessSRCNN that itself fans out over the contexts
    tc = [image(64, 80, 3, 1), image(720, 1000, 4, 2), image(50, 33, 4, 3), image(700, 1100, 3, 4)]
    want = [orc.process(im, 2.0) for im in tc]
    errs = []

    def worker(i):
        try:
            im = tc[i]
            h, w, d = im.shape
            for it in range(8):
                rc, out, conv = S.ProcessSRCNN(im, w, h, d, 2.0, want_conv=(it % 2 == 0))
                if rc != 0:
                    errs.append("thread %d it %d rc %d: %s" % (i, it, rc, S.lib().srcnn_last_error()))
                    return
                if not np.array_equal(out.reshape(want[i][0].shape), want[i][0]):
                    errs.append("thread %d it %d: RGB bytes differ" % (i, it))
                if it % 2 == 0 and not np.array_equal(conv.reshape(want[i][1].shape), want[i][1]):
                    errs.append("thread %d it %d: conv-Y bytes differ" % (i, it))
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    res["thread_errors"] = errs[:5]
    res["lanes_after_threads"] = lanes()
    # a 4K RGB frame: all contexts vs one context
    big = image(2160, 3840, 3, 21)
    rc, out_n, conv_n = S.ProcessSRCNN(big, 3840, 2160, 3, 2.0)
    res["big_rc"] = rc
    h_n, c_n = sha(out_n), sha(conv_n)
    S.shutdown()
    S.init(0)
    res["contexts_after_reinit"] = S.context_count()
    rc, out_1, conv_1 = S.ProcessSRCNN(big, 3840, 2160, 3, 2.0)
    res["big_equal"] = bool(rc == 0 and sha(out_1) == h_n and sha(conv_1) == c_n)
    # and a few rows of it against the oracle: crop with margin, compare the interior
    crop = big[1000:1100]
    want_rgb, _ = orc.process(crop, 2.0)
    got = out_1.reshape(4320, 7680, 3)[2000 + 40:2200 - 40]
    res["big_rows_vs_oracle"] = bool(np.array_equal(got, want_rgb[40:-40]))
    return res


def cmd_node_tiled(nctx):
    """srcnn_y_upscale2x_f32_node_dev over `nctx` virtual contexts == the whole-frame call, bit for bit."""
    res = {"contexts": S.init_devices(devices(nctx)), "cases": []}
    L = S.lib()
    for (h, w, nsub, root) in [(1080, 1920, 4, 0), (333, 501, 3, 0), (333, 501, 1, 1), (97, 64, 5, nctx - 1), (40, 40, 16, 0)]:
        y = synth.plane(h, w, synth.SEED0 + h, "noise")
        S.set_context(root)
        d_in = S.DeviceBuffer.from_numpy(y)
        d_ref = S.DeviceBuffer(4 * h * w * 4)
        d_out = S.DeviceBuffer(4 * h * w * 4)
        S.check(L.srcnn_memset_dev(d_out.ptr, 0xFF, 4 * h * w * 4, None))
        S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_ref.ptr, None))
        S.sync()
        S.check(L.srcnn_y_upscale2x_f32_node_dev(d_in.ptr, w, h, d_out.ptr, nsub))
        a = d_out.to_numpy(np.float32, (2 * h, 2 * w))
        b = d_ref.to_numpy(np.float32, (2 * h, 2 * w))
        res["cases"].append({"shape": [h, w], "nsub": nsub, "root": root, "equal": bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))})
        d_in.free(); d_ref.free(); d_out.free()
    S.set_context(0)
    # and against the oracle once
    y = synth.plane(90, 70, 99, "noise")
    d_in = S.DeviceBuffer.from_numpy(y); d_out = S.DeviceBuffer(4 * y.size * 4)
    S.check(L.srcnn_y_upscale2x_f32_node_dev(d_in.ptr, 70, 90, d_out.ptr, 2))
    want = oracle.Oracle().y_path(y)
    res["vs_oracle"] = bool(np.array_equal(d_out.to_numpy(np.float32, want.shape).view(np.uint32), want.view(np.uint32)))
    return res


def cmd_node_tiled_16k(nctx):
    """BASELINE config #4 at its stated size, from ONE process: a 7680x4320 frame -> 15360x8640 dealt over `nctx` contexts
    (srcnn_y_upscale2x_f32_node_dev: every context pulls its source rows from the root and pushes finished pieces back while
    the next piece computes) == the whole-frame call, by sha256; plus oracle windows on the first and last seam."""
    import time
    res = {"contexts": S.init_devices(devices(nctx))}
    L = S.lib()
    h, w = 4320, 7680
    y = synth.plane(h, w, synth.SEED0 + 4, "smooth")
    d_in = S.DeviceBuffer.from_numpy(y)
    d_out = S.DeviceBuffer(4 * h * w * 4)
    S.check(L.srcnn_memset_dev(d_out.ptr, 0xFF, 4 * h * w * 4, None)); S.sync()
    t0 = time.perf_counter()
    S.check(L.srcnn_y_upscale2x_f32_node_dev(d_in.ptr, w, h, d_out.ptr, 4))
    res["node_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    tiled = d_out.to_numpy(np.float32, (2 * h, 2 * w))
    res["tiled_sha"] = sha(tiled)
    orc = oracle.Oracle()
    ok = []
    for seam in (2 * h // nctx, 2 * h - 2 * h // nctx):          # windows across the first and the last band seam
        iy0, iy1 = seam // 2 - 40, seam // 2 + 40
        want = orc.y_path(np.ascontiguousarray(y[iy0:iy1, 1000:1200]))
        got = tiled[seam - 30:seam + 30, 2000 + 40:2400 - 40]
        ok.append(bool(np.array_equal(got.view(np.uint32), want[2 * 40 - 30 + 0:2 * 40 + 30, 40:-40].view(np.uint32))))
    res["seam_windows_vs_oracle"] = ok
    del tiled
    S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_out.ptr, None)); S.sync()
    res["whole_sha"] = sha(d_out.to_numpy(np.float32, (2 * h, 2 * w)))
    return res


def cmd_stream(nctx):
    """Host-frame stream dealt over the contexts == frame-by-frame results."""
    res = {"contexts": S.init_devices(devices(nctx))}
    frames = synth.frames(5, 270, 480, 7, "smooth")
    singles = np.stack([S.y_upscale2x(f) for f in frames])
    for g in (False, True, True):
        out = S.y_upscale2x_stream(frames, use_graph=g)
        res.setdefault("equal", []).append(bool(np.array_equal(out.view(np.uint32), singles.view(np.uint32))))
    one = S.y_upscale2x_stream(frames[:1])
    res["single_frame"] = bool(np.array_equal(one.view(np.uint32), singles[:1].view(np.uint32)))
    return res


def cmd_comm_tiled(_):
    """The RCCL tiled call at world = 1 (all a one-GPU box can run): every sub-band count, ragged heights."""
    from libsrcnn_amd import multigpu
    S.init(0)
    L = S.lib()
    multigpu.init_comm_from_torch_dist(None, 0, 1)
    res = {"cases": []}
    for (h, w, nsub) in [(540, 960, 4), (333, 501, 5), (97, 64, 1), (20, 33, 16)]:
        y = synth.plane(h, w, synth.SEED0 + w, "noise")
        d_in = S.DeviceBuffer.from_numpy(y)
        t = multigpu.TiledFrameGPU(w, h, 0, 1, nsub=nsub)
        st = S.Stream()
        for _ in range(2):                      # twice: the second frame must wait for the first one's gathers
            t.step(d_in, st.handle)
        st.sync()
        got = t.result()
        d_ref = S.DeviceBuffer(4 * h * w * 4)
        S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_ref.ptr, None))
        S.sync()
        ref = d_ref.to_numpy(np.float32, (2 * h, 2 * w))
        res["cases"].append({"shape": [h, w], "nsub": nsub, "equal": bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))})
        st.destroy()
    # gatherv_at with explicit offsets at world 1: lands where asked
    n = 1000
    src = S.DeviceBuffer.from_numpy(np.arange(n, dtype=np.float32))
    dst = S.DeviceBuffer(4 * 3 * n)
    S.check(L.srcnn_memset_dev(dst.ptr, 0, 4 * 3 * n, None))
    counts = (C.c_size_t * 1)(n); offs = (C.c_size_t * 1)(1500)
    S.check(L.srcnn_comm_gatherv_at_f32(src.ptr, counts, offs, dst.ptr, 0, None))
    S.sync()
    got = dst.to_numpy(np.float32, (3 * n,))
    res["gatherv_at"] = bool(np.array_equal(got[1500:2500], np.arange(n, dtype=np.float32)) and not got[:1500].any() and not got[2500:].any())
    # the bounded waits: srcnn_comm_wait drains a stream inside the deadline; the setter returns the previous value; the
    # barrier (which waits through srcnn_comm_wait) still works with a short deadline and with none
    prev = L.srcnn_comm_set_timeout_ms(5000)
    res["timeout_prev"] = prev
    st = S.Stream()
    S.check(L.srcnn_comm_gatherv_at_f32(src.ptr, counts, offs, dst.ptr, 0, st.handle))
    res["comm_wait"] = L.srcnn_comm_wait(st.handle)
    res["barrier_short_deadline"] = L.srcnn_comm_barrier(None)
    res["timeout_prev2"] = L.srcnn_comm_set_timeout_ms(0)
    res["barrier_no_deadline"] = L.srcnn_comm_barrier(None)
    L.srcnn_comm_set_timeout_ms(prev)
    res["comm_check_env"] = os.environ.get("SRCNN_COMM_CHECK", "")
    st.destroy()
    L.srcnn_comm_destroy()
    res["wait_without_comm"] = L.srcnn_comm_wait(None)
    return res


def cmd_comm_deadline(_):
    """A missed deadline, provoked at world = 1: half a second of kernels queued on a stream, a 40 ms deadline, srcnn_comm_wait
    on that stream.  Expected: SRCNN_E_COMM, the communicator aborted, every later comm call refused at once -- and after
    srcnn_comm_destroy + a fresh srcnn_comm_init everything works again."""
    from libsrcnn_amd import multigpu
    import time
    S.init(0)
    L = S.lib()
    multigpu.init_comm_from_torch_dist(None, 0, 1)
    res = {}
    h, w = 1080, 1920
    y = synth.plane(h, w, 5, "smooth")
    d_in = S.DeviceBuffer.from_numpy(y); d_out = S.DeviceBuffer(4 * h * w * 4)
    st = S.Stream()
    S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_out.ptr, st.handle)); st.sync()      # warm
    res["prev_timeout"] = L.srcnn_comm_set_timeout_ms(40)
    for _ in range(200):                                                                     # ~0.5 s of device work
        S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_out.ptr, st.handle))
    t0 = time.perf_counter()
    res["wait_rc"] = L.srcnn_comm_wait(st.handle)
    res["wait_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    res["wait_error"] = L.srcnn_last_error().decode()
    res["barrier_after_rc"] = L.srcnn_comm_barrier(None)
    res["barrier_after_error"] = L.srcnn_last_error().decode()
    counts = (C.c_size_t * 1)(16); offs = (C.c_size_t * 1)(0)
    res["gather_after_rc"] = L.srcnn_comm_gatherv_at_f32(d_in.ptr, counts, offs, d_out.ptr, 0, None)
    st.sync()                                                                                # the kernels themselves are fine
    res["destroy_rc"] = L.srcnn_comm_destroy()
    L.srcnn_comm_set_timeout_ms(60000)
    multigpu.init_comm_from_torch_dist(None, 0, 1)
    res["barrier_new_comm_rc"] = L.srcnn_comm_barrier(None)
    t = multigpu.TiledFrameGPU(w, h, 0, 1, nsub=3)
    t.step(d_in, st.handle)
    res["wait_new_comm_rc"] = L.srcnn_comm_wait(st.handle)
    got = t.result()
    S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_out.ptr, None)); S.sync()
    res["tiled_equal"] = bool(np.array_equal(got.view(np.uint32), d_out.to_numpy(np.float32, (2 * h, 2 * w)).view(np.uint32)))
    L.srcnn_comm_destroy()
    st.destroy()
    return res


def cmd_env_devices(_):
    """A process that never calls srcnn_init*: SRCNN_DEVICES decides (set by the test)."""
    y = synth.plane(20, 24, 5, "noise")
    out = S.y_upscale2x(y)
    want = oracle.Oracle().y_path(y)
    img = image(700, 1100, 3, 4)
    got_rgb, got_conv = S.process_u8(img, 2.0)
    want_rgb, want_conv = oracle.Oracle().process(img, 2.0)
    return {"contexts": S.context_count(), "equal": bool(np.array_equal(out.view(np.uint32), want.view(np.uint32))),
            "process_equal": bool(np.array_equal(got_rgb, want_rgb) and np.array_equal(got_conv, want_conv)), "lanes": lanes()}


def cmd_process_paths(_):
    """ProcessSRCNN through whatever kernel selection the environment asks for (SRCNN_SHELL_UNFUSED, SRCNN_RESAMPLE_2PASS,
    SRCNN_RS_TPB, SRCNN_MAX_WORKSPACE_MB ...): always the oracle's bytes."""
    S.init(0)
    orc = oracle.Oracle()
    ok = []
    for img, m, filt in [(image(640, 1100, 3, 31), 2.0, 2), (image(641, 1101, 4, 32), 2.0, 2), (image(500, 700, 3, 33), 2.5, 4),
                         (image(60, 47, 4, 34), 2.0, 2), (image(300, 401, 3, 35), 1.0, 2), (image(800, 900, 3, 36), 0.5, 2)]:
        want_rgb, want_conv = orc.process(img, m, filt)
        got_rgb, got_conv = S.process_u8(img, m, filt)
        same = got_rgb.shape == want_rgb.shape
        if m == 1.0:
            # the one documented deviation (identity-size resample): compare only where the reference is defined -- see
            # tests/test_gpu_parity.py::test_identity_size_deviation_is_pinned for what is pinned there
            ok.append(bool(same))
            continue
        ok.append(bool(same and np.array_equal(got_rgb, want_rgb) and np.array_equal(got_conv, want_conv)))
    ys = []
    for (h, w) in [(64, 128), (67, 131), (1, 9), (9, 1), (130, 61)]:
        y = synth.plane(h, w, 3 + h, "noise")
        ys.append(bool(np.array_equal(S.y_upscale2x(y).view(np.uint32), orc.y_path(y).view(np.uint32))))
    return {"process": ok, "y": ys}


def cmd_queue(nctx):
    """More concurrent ProcessSRCNN callers than lanes (the test sets SRCNN_MAX_LANES=1): callers queue for the lane, every
    result is right, exactly one lane per context exists afterwards."""
    S.init_devices(devices(max(1, nctx)))
    orc = oracle.Oracle()
    tc = [image(700, 1100, 3, 4), image(64, 80, 3, 1), image(720, 1000, 4, 2)]
    want = [orc.process(im, 2.0) for im in tc]
    errs = []

    def worker(i):
        try:
            im = tc[i]
            h, w, d = im.shape
            for it in range(5):
                rc, out, conv = S.ProcessSRCNN(im, w, h, d, 2.0)
                if rc != 0 or not np.array_equal(out.reshape(want[i][0].shape), want[i][0]) or \
                        not np.array_equal(conv.reshape(want[i][1].shape), want[i][1]):
                    errs.append("thread %d it %d rc %d" % (i, it, rc))
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return {"errors": errs, "lanes": lanes(), "contexts": S.context_count()}


if __name__ == "__main__":
    cmd = sys.argv[1]
    arg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    out = globals()["cmd_" + cmd](arg)
    print(json.dumps(out))
