"""GPU: the BASELINE.json configurations at their full sizes, the re-entrancy of the drop-in boundary, guard bands
around every device output, and the lifetime rules of tables / graphs / workspaces.

Full-size frames cannot be pushed through the CPU oracle whole (the reference needs ~100 planes per frame), so
they are checked through oracle WINDOWS: the oracle is run on an input crop whose margin (16 input px) is far wider
than the path's receptive field (+-6 output px = +-5 input px incl. the 4-tap resampler), and the interior of
its output must equal the same window of the GPU frame bit for bit.  Windows are placed on both true borders, on
every band seam, and at seeded random interior positions.
"""
import ctypes as C
import hashlib
import threading

import numpy as np
import pytest

from conftest import assert_bit_equal
from libsrcnn_amd import multigpu, synth

pytestmark = pytest.mark.gpu

PAD = 16      # input-pixel margin of an oracle crop


def oracle_window(oracle_lib, y, oy, ox, wh, ww):
    """Oracle value of output window [oy,oy+wh) x [ox,ox+ww) of the 2x frame of `y` (any position, borders
    included): crop the input with PAD px of margin where the frame has them, run the oracle, cut the window."""
    h, w = y.shape
    iy0, iy1 = max(0, oy // 2 - PAD), min(h, (oy + wh + 1) // 2 + PAD)
    ix0, ix1 = max(0, ox // 2 - PAD), min(w, (ox + ww + 1) // 2 + PAD)
    out = oracle_lib.y_path(np.ascontiguousarray(y[iy0:iy1, ix0:ix1]))
    return out[oy - 2 * iy0: oy - 2 * iy0 + wh, ox - 2 * ix0: ox - 2 * ix0 + ww]


def check_windows(oracle_lib, y, frame, windows, what):
    for oy, ox, wh, ww in windows:
        assert_bit_equal(frame[oy:oy + wh, ox:ox + ww], oracle_window(oracle_lib, y, oy, ox, wh, ww),
                         "%s window %dx%d at (%d,%d)" % (what, wh, ww, oy, ox))


# Window positions rotate from run to run (the fixed borders / seams stay): the per-run salt is printed once so that a
# failure can be replayed with SRCNN_TEST_SEED=<salt>.
from conftest import rotating_seed


def run_salt():
    return rotating_seed("oracle-window positions of the full-size configuration tests")


def frame_windows(out_h, out_w, seams, seed, n_random=2):
    rng = np.random.default_rng([seed, run_salt()])
    wins = [(0, 0, 40, 64), (out_h - 40, out_w - 64, 40, 64), (0, out_w - 64, 24, 64), (out_h - 24, 0, 24, 64)]
    for s in seams:                       # a window straddling each seam, at a seeded x
        ox = int(rng.integers(0, out_w - 96))
        wins.append((s - 20, ox, 40, 96))
    for _ in range(n_random):
        wins.append((int(rng.integers(50, out_h - 100)), int(rng.integers(50, out_w - 200)), 48, 96))
    return wins


# ------------------------------------------------------------------------------------------------
# config #4: one 7680x4320 frame -> 15360x8640, tiled into 8 bands (+ the automatic banding of the whole-frame call)
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def frame8k():
    return synth.plane(4320, 7680, synth.SEED0 + 4, "smooth")


def test_config4_8k_frame_as_8_bands(srcnn, oracle_lib, frame8k):
    S, L = srcnn, srcnn.lib()
    y = frame8k
    h, w = y.shape
    OH, OW = 2 * h, 2 * w
    d_in = S.DeviceBuffer.from_numpy(y)
    d_out = S.DeviceBuffer(OH * OW * 4)
    S.check(L.srcnn_memset_dev(d_out.ptr, 0xFF, OH * OW * 4, None))
    seams = []
    for r in range(8):                    # the 8 bands a node's 8 ranks would own, here one after the other
        row0, rows = multigpu.band_rows(OH, r, 8)
        assert rows == 1080
        if row0:
            seams.append(row0)
        S.check(L.srcnn_y_upscale2x_f32_band_dev(d_in.ptr, w, h, row0, rows, d_out.ptr + row0 * OW * 4, None))
    S.sync()
    bands = d_out.to_numpy(np.float32, (OH, OW))
    check_windows(oracle_lib, y, bands, frame_windows(OH, OW, seams, seed=41), "8 bands")
    sha_bands = hashlib.sha256(bands.tobytes()).hexdigest()
    del bands

    # the whole-frame call with the scratch limit at 4 GiB (the layer-2 planes of this frame are 17 GB): produced in
    # sub-bands of 2180 rows internally
    prev = L.srcnn_set_workspace_limit(4 << 30)
    try:
        S.check(L.srcnn_memset_dev(d_out.ptr, 0xFF, OH * OW * 4, None))
        S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_out.ptr, None))
        S.sync()
    finally:
        L.srcnn_set_workspace_limit(prev)
    whole = d_out.to_numpy(np.float32, (OH, OW))
    assert hashlib.sha256(whole.tobytes()).hexdigest() == sha_bands       # tiled == whole frame, bit for bit
    sub = (4 << 30) // (128 * OW) - 4                                      # rows per internal sub-band
    assert 3 * sub < OH
    check_windows(oracle_lib, y, whole, frame_windows(OH, OW, [sub, 2 * sub, 3 * sub], seed=43, n_random=1),
                  "auto-banded whole frame")


def test_config4_tiled_driver_with_rccl_gather(srcnn, oracle_lib):
    """multigpu.TiledFrameGPU = band compute + srcnn_comm_gatherv_f32, as bench.py --workload tiled8k runs it,
    with a real (1-rank) RCCL communicator: assembled frame == whole-frame call == oracle windows."""
    S, L = srcnn, srcnn.lib()
    h, w = 1081, 1922                    # odd output geometry on purpose
    y = synth.plane(h, w, synth.SEED0 + 44, "noise")
    multigpu.init_comm_from_torch_dist(None, 0, 1)
    try:
        rank, world = C.c_int(-1), C.c_int(-1)
        S.check(L.srcnn_comm_rank(C.byref(rank), C.byref(world)))
        assert (rank.value, world.value) == (0, 1)
        d_in = S.DeviceBuffer.from_numpy(y)
        t = multigpu.TiledFrameGPU(w, h, 0, 1)
        t.step(d_in)
        S.sync()
        got = t.result()
    finally:
        S.check(L.srcnn_comm_destroy())
    assert_bit_equal(got, S.y_upscale2x(y), "tiled driver vs whole frame")
    check_windows(oracle_lib, y, got, frame_windows(2 * h, 2 * w, [], seed=5), "tiled driver")


def test_gatherv_places_ragged_counts(srcnn):
    """Per-rank counts: with one rank the root's own band must land at offset 0 and nothing else be touched;
    a zero count is legal.  (N>1 placement = the same prefix sums; host logic in tests/test_multi_gpu_cpu.py.)"""
    S, L = srcnn, srcnn.lib()
    multigpu.init_comm_from_torch_dist(None, 0, 1)
    try:
        x = np.arange(1, 1001, dtype=np.float32)
        src = S.DeviceBuffer.from_numpy(x)
        dst = S.DeviceBuffer(4 * 1200)
        S.check(L.srcnn_memset_dev(dst.ptr, 0, 4 * 1200, None))
        counts = (C.c_size_t * 1)(777)
        S.check(L.srcnn_comm_gatherv_f32(src.ptr, counts, dst.ptr, 0, None))
        S.sync()
        got = dst.to_numpy(np.float32, (1200,))
        assert np.array_equal(got[:777], x[:777]) and not got[777:].any()
        counts[0] = 0
        S.check(L.srcnn_comm_gatherv_f32(None, counts, dst.ptr, 0, None))
        assert L.srcnn_comm_gatherv_f32(src.ptr, counts, dst.ptr, 3, None) == -204      # bad root
    finally:
        S.check(L.srcnn_comm_destroy())
    assert L.srcnn_comm_gatherv_f32(src.ptr, counts, dst.ptr, 0, None) == -204          # no communicator


# ------------------------------------------------------------------------------------------------
# config #3: batch of 64 resident 1080p frames; config #5: stream of 4K frames with per-slot hipGraph
# ------------------------------------------------------------------------------------------------
def test_config3_batch_of_64_1080p_vs_oracle(srcnn, oracle_lib):
    S, L = srcnn, srcnn.lib()
    h, w, F = 1080, 1920, 64
    d_in = S.DeviceBuffer(F * h * w * 4)
    d_out = S.DeviceBuffer(F * 4 * h * w * 4)
    keep = {}
    for f in range(F):
        y = synth.plane(h, w, synth.SEED0 + f, "smooth" if f % 2 else "noise")
        d_in.upload(y, offset=f * h * w * 4)
        if f in (0, 31, 63):
            keep[f] = y
    S.check(L.srcnn_memset_dev(d_out.ptr, 0xFF, F * 4 * h * w * 4, None))
    S.check(L.srcnn_y_upscale2x_f32_batch_dev(d_in.ptr, w, h, F, d_out.ptr, None))
    S.sync()
    for f, y in keep.items():
        got = d_out.to_numpy(np.float32, (2 * h, 2 * w), offset=f * 4 * h * w * 4)
        check_windows(oracle_lib, y, got, frame_windows(2 * h, 2 * w, [], seed=300 + f), "batch frame %d" % f)
    # every frame was written (no 0xFF.. NaN pattern left) and frames differ from each other
    probe = [d_out.to_numpy(np.float32, (2 * w,), offset=f * 4 * h * w * 4 + 4 * (2 * w) * 1000) for f in range(F)]
    assert not np.isnan(np.stack(probe)).any()
    assert len({p.tobytes() for p in probe}) == F


@pytest.mark.parametrize("use_graph", [True, False])
def test_config5_stream_of_4k_frames(srcnn, oracle_lib, use_graph):
    S = srcnn
    h, w, F = 2160, 3840, 8
    fr = np.stack([synth.plane(h, w, synth.SEED0 + 500 + f, "noise" if f % 3 == 0 else "smooth") for f in range(F)])
    got = S.y_upscale2x_stream(fr, use_graph=use_graph)
    rng = np.random.default_rng(77)
    for f in range(F):
        oy, ox = int(rng.integers(0, 2 * h - 48)), int(rng.integers(0, 2 * w - 96))
        check_windows(oracle_lib, fr[f], got[f], [(oy, ox, 48, 96)], "stream frame %d (graph=%s)" % (f, use_graph))
    for f in (0, 1, 7):                  # slot 0 eager, slot 1 eager, a replayed frame: all equal the single-frame call
        assert_bit_equal(got[f], S.y_upscale2x(fr[f]), "stream frame %d vs single call" % f)


def test_bands_equal_oracle_crops(srcnn, oracle_lib):
    """Bands against the ORACLE's rows (not against another HIP run), incl. bands inside a border's receptive field."""
    y = synth.plane(45, 70, synth.SEED0 + 77, "noise")
    want = oracle_lib.y_path(y)
    for row0, rows in ((0, 90), (0, 11), (11, 23), (34, 1), (35, 55), (88, 2), (3, 5), (89, 1)):
        assert_bit_equal(srcnn.y_upscale2x_band(y, row0, rows), want[row0:row0 + rows], "band %d+%d" % (row0, rows))
    for world in (3, 7, 8):              # ragged partitions: 90 rows over 7 ranks = 13,13,13,13,13,13,12
        parts = [srcnn.y_upscale2x_band(y, *multigpu.band_rows(90, r, world)) for r in range(world)]
        assert_bit_equal(np.concatenate(parts), want, "%d ragged bands" % world)


# ------------------------------------------------------------------------------------------------
# fast tiers: tighter bound, full-size windows
# ------------------------------------------------------------------------------------------------
# Measured against the reference on whole 1080p / 4K frames of both generators (tools/fast_error.py, bench.py):
# fp32-FMA tier <= 2.8e-4, split-fp16 tier <= 4.3e-4.  The reference itself is 2.3e-4 away from exact arithmetic
# (SURVEY.md 7), so no evaluation that differs from its rounding order can be pinned much closer than this.
TOL_FAST = 5e-4
TOL_FAST_F16 = 6e-4
TOL = {"MODE_FAST": TOL_FAST, "MODE_FAST_F16": TOL_FAST_F16}


@pytest.mark.parametrize("mode_name", ["MODE_FAST", "MODE_FAST_F16"])
def test_fast_tiers_1080p_border_and_interior(srcnn, oracle_lib, mode_name):
    S = srcnn
    h, w = 1080, 1920
    worst = 0.0
    for kind in ("smooth", "noise"):
        y = synth.plane(h, w, synth.SEED0 + 61, kind)
        prev = S.set_mode(getattr(S, mode_name))
        try:
            got = S.y_upscale2x(y)
        finally:
            S.set_mode(prev)
        for oy, ox, wh, ww in frame_windows(2 * h, 2 * w, [], seed=9, n_random=3):
            want = oracle_window(oracle_lib, y, oy, ox, wh, ww)
            err = float(np.max(np.abs(got[oy:oy + wh, ox:ox + ww].astype(np.float64) - want)))
            worst = max(worst, err)
            assert err <= TOL[mode_name], (mode_name, kind, (oy, ox), err)
    assert worst > 0.0


def test_fused_f16_bands_tiles_and_chunks(srcnn, oracle_lib):
    """The fused split-fp16 kernel (one launch for all three layers): frames whose width/height straddle its 60-column
    wave strips, 480-column workgroup strips and row chunks; horizontal bands; and run-to-run determinism."""
    S = srcnn
    prev = S.set_mode(S.MODE_FAST_F16)
    try:
        for h, w, kind in ((37, 29, "noise"), (40, 31, "smooth"), (100, 245, "noise"), (130, 270, "smooth"), (9, 500, "noise")):
            y = synth.plane(h, w, synth.SEED0 + 7 * h + w, kind)
            want = oracle_lib.y_path(y).astype(np.float64)
            got = S.y_upscale2x(y)
            assert float(np.max(np.abs(got - want))) <= TOL_FAST_F16, (h, w)
            assert np.array_equal(got.view(np.uint32), S.y_upscale2x(y).view(np.uint32)), "non-deterministic"
            for row0, rows in ((0, 1), (2 * h - 1, 1), (3, 7), (2 * h - 9, 9), (0, 2 * h)):
                band = S.y_upscale2x_band(y, row0, rows)
                # a band is the same arithmetic on the same values, whatever chunk it falls into
                assert np.array_equal(band.view(np.uint32), got[row0:row0 + rows].view(np.uint32)), (h, w, row0, rows)
    finally:
        S.set_mode(prev)


def test_fused_f16_strip_and_chunk_edges_sweep(srcnn, oracle_lib):
    """Seeded sweep of frame shapes whose 2x outputs straddle the fused kernel's decomposition: 60-column wave strips,
    480-column workgroup strips, 4-row ring stages, 16-row minimum chunks and the 1-/2-row degenerate cases."""
    S = srcnn
    rng = np.random.default_rng(2026)
    widths = [14, 15, 16, 29, 30, 31, 119, 120, 121, 239, 240, 241, 245]
    heights = [1, 2, 3, 7, 8, 9, 17, 33]
    shapes = [(int(rng.choice(heights)), int(rng.choice(widths))) for _ in range(16)] + [(1, 241), (33, 1), (2, 30)]
    prev = S.set_mode(S.MODE_FAST_F16)
    try:
        for k, (h, w) in enumerate(shapes):
            y = synth.plane(h, w, synth.SEED0 + 700 + k, "noise" if k % 2 else "smooth")
            err = float(np.max(np.abs(S.y_upscale2x(y).astype(np.float64) - oracle_lib.y_path(y))))
            assert err <= TOL_FAST_F16, (h, w, err)
    finally:
        S.set_mode(prev)


def test_fast_tiers_through_processsrcnn_differ_by_at_most_one_level(srcnn, oracle_lib):
    """What a ProcessSRCNN user sees when opting into a non-parity tier: the u8 image equals the reference's except where
    a |dY| of a few 1e-4 moves a value across an integer boundary before truncation -- never by more than one level, and
    in well under 1 % of the bytes (strict mode: zero bytes)."""
    S = srcnn
    img = _image(300, 420, 3, 77)
    want_rgb, want_conv = oracle_lib.process(img, 2.0)
    S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
    for mode in (S.MODE_FAST, S.MODE_FAST_F16):
        prev = S.set_mode(mode)
        try:
            rc, out, conv = S.ProcessSRCNN(img, 420, 300, 3, 2.0)
        finally:
            S.set_mode(prev)
        assert rc == 0
        for got, want in ((out.reshape(want_rgb.shape), want_rgb), (conv.reshape(want_conv.shape), want_conv)):
            d = np.abs(got.astype(np.int16) - want.astype(np.int16))
            assert int(d.max()) <= 1, (mode, int(d.max()))
            assert float((d > 0).mean()) < 0.01, (mode, float((d > 0).mean()))
    rc, out, conv = S.ProcessSRCNN(img, 420, 300, 3, 2.0)
    assert np.array_equal(out.reshape(want_rgb.shape), want_rgb) and np.array_equal(conv.reshape(want_conv.shape), want_conv)


def test_f16_tier_in_a_fresh_process(oracle_lib, tmp_path):
    """The fused fp16 tier (the only form since round 5: the two-kernel form k_conv12_f16 + k_conv3_fast is gone) from a
    process of its own, inside its documented bound."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, libsrcnn_amd as S, oracle;"
            "from libsrcnn_amd import synth; S.init(0); S.set_mode(S.MODE_FAST_F16);"
            "y = synth.plane(70, 150, 11, 'noise');"
            "e = float(np.max(np.abs(S.y_upscale2x(y).astype(np.float64) - oracle.Oracle().y_path(y)))); print('ERR', e)") % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True,
                       timeout=300)
    if "strict-only build" in r.stderr:
        pytest.skip("strict-only build: the fp16 tier is not compiled in")
    assert "ERR" in r.stdout, r.stdout + r.stderr
    err = float(r.stdout.split("ERR")[1])
    assert 0.0 < err <= TOL_FAST_F16, err


# ------------------------------------------------------------------------------------------------
# the drop-in boundary is re-entrant like the reference's
# ------------------------------------------------------------------------------------------------
def _image(h, w, d, seed):
    rng = np.random.default_rng(seed)
    base = synth.plane(h, w, synth.SEED0 + seed, "smooth")
    img = np.empty((h, w, d), np.uint8)
    for k in range(d):
        img[..., k] = np.clip(base * (0.55 + 0.15 * k) + rng.integers(0, 40, base.shape), 0, 255).astype(np.uint8)
    return img


def test_processsrcnn_concurrent_host_threads(srcnn, oracle_lib):
    """4 host threads call ProcessSRCNN at once -- small images and images above the 8 MB pipelined threshold,
    RGB and RGBA -- 20 iterations each; every result must equal the oracle's bytes."""
    S = srcnn
    S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
    cases = [_image(64, 80, 3, 1), _image(720, 1000, 4, 2), _image(50, 33, 4, 3), _image(700, 1100, 3, 4)]
    want = [oracle_lib.process(img, 2.0) for img in cases]
    assert cases[1].nbytes * 4 >= (8 << 20) and cases[3].nbytes * 4 >= (8 << 20) and cases[0].nbytes * 4 < (8 << 20)
    errs = []

    def worker(i):
        try:
            img = cases[i]
            h, w, d = img.shape
            for it in range(20):
                rc, out, conv = S.ProcessSRCNN(img, w, h, d, 2.0, want_conv=(it % 2 == 0))
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
    assert not errs, errs[:5]
    tables, lanes = C.c_int(), C.c_int()
    S.lib().srcnn_debug_counts(C.byref(tables), C.byref(lanes))
    assert 1 <= lanes.value <= 4


def test_mixed_entry_points_concurrently(srcnn, oracle_lib):
    """ProcessSRCNN, a _dev call on the NULL stream and a _dev call on a private stream from three threads."""
    S = srcnn
    img = _image(300, 420, 3, 9)
    want_rgb, _ = oracle_lib.process(img, 2.0)
    ys = [synth.plane(60, 90, synth.SEED0 + 70 + i, "noise") for i in range(2)]
    want_y = [oracle_lib.y_path(y) for y in ys]
    errs = []

    def proc():
        for _ in range(10):
            rc, out, _c = S.ProcessSRCNN(img, 420, 300, 3, 2.0, want_conv=False)
            if rc or not np.array_equal(out.reshape(want_rgb.shape), want_rgb):
                errs.append("ProcessSRCNN mismatch rc=%d" % rc)

    def dev(i, own_stream):
        st = S.Stream() if own_stream else None
        din = S.DeviceBuffer.from_numpy(ys[i])
        dout = S.DeviceBuffer(want_y[i].nbytes)
        for _ in range(10):
            S.check(S.lib().srcnn_y_upscale2x_f32_dev(din.ptr, 90, 60, dout.ptr, st.handle if st else None))
            st.sync() if st else S.check(S.lib().srcnn_stream_sync(None))
            if not np.array_equal(dout.to_numpy(np.float32, want_y[i].shape).view(np.uint32), want_y[i].view(np.uint32)):
                errs.append("dev call %d mismatch" % i)
        if st:
            st.destroy()

    ts = [threading.Thread(target=proc), threading.Thread(target=dev, args=(0, False)), threading.Thread(target=dev, args=(1, True))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs[:5]


# ------------------------------------------------------------------------------------------------
# lifetime rules: table cache churn, graphs vs. later work on the same stream
# ------------------------------------------------------------------------------------------------
def test_table_cache_churn_never_invalidates(srcnn, oracle_lib):
    """More distinct frame sizes than the cache bound (64 tables = 32 non-square sizes): every result stays equal
    to the oracle, the cache stays bounded, and a graph captured BEFORE the churn (its tables long evicted from
    the cache) and a larger eager frame on the graph's stream (which re-allocates that stream's own scratch) do
    not disturb the graph's replay."""
    S, L = srcnn, srcnn.lib()
    fr = synth.frames(2, 30, 46, 900, "noise")
    want = np.stack([oracle_lib.y_path(f) for f in fr])
    din = S.DeviceBuffer.from_numpy(fr)
    dout = S.DeviceBuffer(want.nbytes)
    st = S.Stream()
    gh = C.c_void_p()
    S.check(L.srcnn_batch_graph_create(din.ptr, 46, 30, 2, dout.ptr, st.handle, C.byref(gh)))

    for k in range(40):                                   # 80 new tables
        h, w = 9 + k, 50 + 2 * k
        y = synth.plane(h, w, synth.SEED0 + 1000 + k, "smooth")
        assert_bit_equal(S.y_upscale2x(y), oracle_lib.y_path(y), "churn size %dx%d" % (h, w))
    tables = C.c_int()
    L.srcnn_debug_counts(C.byref(tables), None)
    assert tables.value <= 64 + 2, tables.value

    big = synth.plane(200, 300, synth.SEED0 + 2000, "noise")           # larger eager frame on the graph's stream
    dbig_in = S.DeviceBuffer.from_numpy(big)
    dbig_out = S.DeviceBuffer(big.nbytes * 4)
    S.check(L.srcnn_y_upscale2x_f32_dev(dbig_in.ptr, 300, 200, dbig_out.ptr, st.handle))
    st.sync()
    assert_bit_equal(dbig_out.to_numpy(np.float32, (400, 600)), oracle_lib.y_path(big), "eager frame on the graph's stream")

    S.check(L.srcnn_memset_dev(dout.ptr, 0, want.nbytes, st.handle))
    for _ in range(2):
        S.check(L.srcnn_batch_graph_launch(gh))
    st.sync()
    assert_bit_equal(dout.to_numpy(np.float32, want.shape), want, "graph replay after churn")
    S.check(L.srcnn_batch_graph_destroy(gh))
    st.destroy()


def test_mode_is_sampled_per_call_and_profiling_survives_capture(srcnn, oracle_lib):
    """A graph captured while profiling is on leaves profiling on (the capture no longer flips a global), and the
    stage timers still count eager launches afterwards."""
    S, L = srcnn, srcnn.lib()
    fr = synth.frames(2, 24, 40, 950, "smooth")
    din = S.DeviceBuffer.from_numpy(fr)
    dout = S.DeviceBuffer(fr.nbytes * 4)
    st = S.Stream()
    S.profile_reset()
    S.profile_enable(True)
    try:
        gh = C.c_void_p()
        S.check(L.srcnn_batch_graph_create(din.ptr, 40, 24, 2, dout.ptr, st.handle, C.byref(gh)))
        assert S.profile_enable(True) == 1                       # still on
        n0 = S.profile_read()["conv12"][1]                       # the eager pre-run was timed: 2 frames
        assert n0 == 2
        S.check(L.srcnn_batch_graph_launch(gh))
        st.sync()
        assert S.profile_read()["conv12"][1] == n0               # replay adds no spans
        S.check(L.srcnn_y_upscale2x_f32_batch_dev(din.ptr, 40, 24, 2, dout.ptr, st.handle))
        st.sync()
        assert S.profile_read()["conv12"][1] == n0 + 2
        S.check(L.srcnn_batch_graph_destroy(gh))
    finally:
        S.profile_enable(False)
        st.destroy()
    assert_bit_equal(dout.to_numpy(np.float32, (2, 48, 80)), np.stack([oracle_lib.y_path(f) for f in fr]), "batch")


# ------------------------------------------------------------------------------------------------
# guard bands: no entry point writes a byte outside its output (the reference has latent bugs of exactly this
# class: src/frawscale.cpp:185-193,249)
# ------------------------------------------------------------------------------------------------
CANARY = 4096


class Guarded:
    """Device buffer of `nbytes` with CANARY bytes of 0xA5 on either side."""

    def __init__(self, S, nbytes):
        self.S, self.nbytes = S, nbytes
        self.buf = S.DeviceBuffer(nbytes + 2 * CANARY)
        S.check(S.lib().srcnn_memset_dev(self.buf.ptr, 0xA5, nbytes + 2 * CANARY, None))
        S.sync()
        self.ptr = self.buf.ptr + CANARY

    def intact(self):
        raw = self.buf.to_numpy(np.uint8, (self.nbytes + 2 * CANARY,))
        return bool((raw[:CANARY] == 0xA5).all() and (raw[-CANARY:] == 0xA5).all())

    def data(self, dtype, shape):
        return self.buf.to_numpy(dtype, shape, offset=CANARY)


ODD_SHAPES = [(1, 1), (1, 17), (13, 1), (67, 131), (5, 64), (33, 3)]


@pytest.mark.parametrize("shape", ODD_SHAPES)
def test_canaries_around_device_outputs(srcnn, oracle_lib, shape):
    S, L = srcnn, srcnn.lib()
    h, w = shape
    y = synth.plane(h, w, synth.SEED0 + 31 * h + w, "noise")
    want, up, c1, c2 = oracle_lib.y_path(y, taps=True)
    d_in = S.DeviceBuffer.from_numpy(y)
    n_out = 4 * h * w

    g = Guarded(S, n_out * 4)                                            # whole frame
    S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, g.ptr, None)); S.sync()
    assert g.intact(), "y_upscale2x wrote outside its output"
    assert_bit_equal(g.data(np.float32, (2 * h, 2 * w)), want, "guarded whole frame")

    for row0, rows in ((2 * h - 1, 1), (0, 1), (max(0, 2 * h - 3), min(3, 2 * h))):      # bands incl. the last row
        gb = Guarded(S, rows * 2 * w * 4)
        S.check(L.srcnn_y_upscale2x_f32_band_dev(d_in.ptr, w, h, row0, rows, gb.ptr, None)); S.sync()
        assert gb.intact(), "band %d+%d wrote outside its output" % (row0, rows)
        assert_bit_equal(gb.data(np.float32, (rows, 2 * w)), want[row0:row0 + rows], "guarded band")

    gr = Guarded(S, n_out * 4)                                           # stage-level entry points
    S.check(L.srcnn_resample_f32_dev(d_in.ptr, w, h, 2 * w, 2 * h, 2, gr.ptr, None)); S.sync()
    assert gr.intact()
    assert_bit_equal(gr.data(np.float32, (2 * h, 2 * w)), up, "guarded resample")
    d_up = S.DeviceBuffer.from_numpy(up)
    g12 = Guarded(S, 32 * n_out * 4)
    S.check(L.srcnn_conv12_f32_dev(d_up.ptr, 2 * w, 2 * h, g12.ptr, None)); S.sync()
    assert g12.intact()
    assert_bit_equal(g12.data(np.float32, (32, 2 * h, 2 * w)), c2, "guarded conv12")
    g1 = Guarded(S, 64 * n_out * 4)
    S.check(L.srcnn_conv1_f32_dev(d_up.ptr, 2 * w, 2 * h, g1.ptr, None)); S.sync()
    assert g1.intact()
    assert_bit_equal(g1.data(np.float32, (64, 2 * h, 2 * w)), c1, "guarded conv1")
    g2 = Guarded(S, 32 * n_out * 4)
    S.check(L.srcnn_conv2_f32_dev(g1.ptr, 2 * w, 2 * h, g2.ptr, None)); S.sync()
    assert g2.intact()
    g3 = Guarded(S, n_out * 4)
    S.check(L.srcnn_conv3_f32_dev(g2.ptr, 2 * w, 2 * h, g3.ptr, None)); S.sync()
    assert g3.intact()
    assert_bit_equal(g3.data(np.float32, (2 * h, 2 * w)), want, "guarded conv3")

    for mode in (S.MODE_FAST, S.MODE_FAST_F16):                          # the non-parity kernels have their own tiles
        prev = S.set_mode(mode)
        try:
            gf = Guarded(S, n_out * 4)
            S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, gf.ptr, None)); S.sync()
        finally:
            S.set_mode(prev)
        assert gf.intact(), "mode %d wrote outside its output" % mode
        assert float(np.max(np.abs(gf.data(np.float32, (2 * h, 2 * w)).astype(np.float64) - want))) <= TOL_FAST_F16


@pytest.mark.parametrize("shape,d,mul", [((1, 1), 3, 2.0), ((1, 17), 4, 2.0), ((13, 1), 3, 2.0), ((67, 131), 4, 2.0),
                                         ((67, 131), 3, 1.5), ((40, 31), 3, 3.0)])
def test_canaries_around_processsrcnn_host_outputs(srcnn, oracle_lib, shape, d, mul):
    """srcnn_process_u8 writes into caller-allocated HOST buffers (out, conv_opt): guard bands on both."""
    S, L = srcnn, srcnn.lib()
    h, w = shape
    img = _image(h, w, d, 7 * h + w)
    want_rgb, want_conv = oracle_lib.process(img, mul)
    dh, dw = want_conv.shape
    out = np.full(dh * dw * d + 2 * CANARY, 0xA5, np.uint8)
    conv = np.full(dh * dw + 2 * CANARY, 0xA5, np.uint8)
    S.check(L.srcnn_process_u8(img.ctypes.data, w, h, d, float(np.float32(mul)), 2, out.ctypes.data + CANARY,
                               conv.ctypes.data + CANARY))
    for buf in (out, conv):
        assert (buf[:CANARY] == 0xA5).all() and (buf[-CANARY:] == 0xA5).all()
    assert np.array_equal(out[CANARY:-CANARY].reshape(want_rgb.shape), want_rgb)
    assert np.array_equal(conv[CANARY:-CANARY].reshape(want_conv.shape), want_conv)


@pytest.mark.gpu
def test_stream_use_graph_auto_is_cheap_in_host_cpu_by_default(srcnn):
    """srcnn_y_upscale2x_f32_stream(use_graph = 1) keeps hipGraph replay only while it is cheap.  On ROCm 7.2
    a replay pins a runtime thread for the frame's duration (~10 ms of CPU per 4K frame, 0.6 ms for plain launches): the
    first call measures its first four replays and falls back; from then on the shape runs on plain launches and costs
    < 1 ms of process CPU per frame.  Whatever the runtime does, what ran is reported, the results are the single-frame
    results, and use_graph = 2 still replays every frame but the slots' first ones."""
    import ctypes as C
    import time
    S = srcnn
    L = S.lib()
    w, h, F = 3840, 2160, 12
    pin_in = S.PinnedArray((F, h, w), np.float32)
    pin_out = S.PinnedArray((F, 2 * h, 2 * w), np.float32)
    try:
        two = synth.frames(2, h, w, 7, "smooth")
        for f in range(F):
            pin_in.array[f] = two[f & 1]
        want = [S.y_upscale2x(two[0]), S.y_upscale2x(two[1])]

        def run(g):
            c0, t0 = time.process_time(), time.perf_counter()
            S.check(L.srcnn_y_upscale2x_f32_stream(pin_in.array.ctypes.data, w, h, F, pin_out.array.ctypes.data, g))
            return (time.process_time() - c0) / F, (time.perf_counter() - t0) / F, S.stream_mode()
        cpu1, wall1, (g1, p1, fell1) = run(1)                      # first auto call: eager frames, capture, probe, verdict
        assert g1 + p1 == F
        for f in (0, 5, F - 1):
            assert_bit_equal(pin_out.array[f], want[f & 1], "auto, frame %d" % f)
        cpu2, wall2, (g2, p2, fell2) = run(1)                      # the verdict is remembered
        assert g2 + p2 == F
        cpu2b, wall2b, mode2b = run(1)                             # (the better of two: process CPU time is everything the process does)
        assert mode2b[:2] == (g2, p2)
        if cpu2b < cpu2:
            cpu2, wall2 = cpu2b, wall2b
        if fell1:
            assert g1 <= 5 and (g2, p2) == (0, F), ((g1, p1), (g2, p2))
        else:
            assert g2 >= F - 2                                     # a runtime whose replay is cheap keeps its graphs
        assert cpu2 < 1e-3, "use_graph=1 costs %.2f ms of host CPU per frame (wall %.2f ms; mode %s)" % (cpu2 * 1e3, wall2 * 1e3, (g2, p2, fell2))
        cpu3, wall3, (g3, p3, fell3) = run(2)                      # forced: every frame but the slots' first (eager) ones
        assert g3 >= F - 2 and not fell3, (g3, p3, fell3)
        for f in (1, F - 1):
            assert_bit_equal(pin_out.array[f], want[f & 1], "forced graph, frame %d" % f)
        print("use_graph auto: first call %s cpu %.2f ms/frame, second %s cpu %.2f ms/frame (wall %.2f); forced graph cpu %.2f ms/frame (wall %.2f)"
              % ((g1, p1, fell1), cpu1 * 1e3, (g2, p2, fell2), cpu2 * 1e3, wall2 * 1e3, cpu3 * 1e3, wall3 * 1e3))
    finally:
        pin_in.free(); pin_out.free()


@pytest.mark.gpu
def test_pageable_host_memory_is_bounced_and_survives_heap_trims(srcnn, oracle_lib):
    """Round 6: pageable host memory never reaches a HIP copy (the runtime pins it in place and a heap trim then invalidates the
    mapping under the copy engine: "Memory access fault by GPU ... <heap address>", docs/HISTORY.md 11).  (a) the plumbing copies
    round-trip pageable arrays of awkward sizes and offsets through the 16 MB bounce slots bit for bit; (b) the host-pointer call on
    heap-resident arrays keeps working -- and stays bit-exact -- while the heap is grown, freed and trimmed between the calls."""
    import ctypes as C
    S = srcnn
    rng = np.random.default_rng(606)
    for nbytes in (1, 4097, (16 << 20) - 3, (16 << 20) + 5, (48 << 20) + 12345):
        src = rng.integers(0, 256, nbytes + 64, dtype=np.uint8)[13:13 + nbytes]          # unaligned, pageable
        d = S.DeviceBuffer(nbytes + 256)
        S.check(S.lib().srcnn_memcpy_h2d(d.ptr + 128, src.ctypes.data, nbytes, None))
        back = np.zeros(nbytes + 7, np.uint8)[7:]
        S.check(S.lib().srcnn_memcpy_d2h(back.ctypes.data, d.ptr + 128, nbytes, None))
        assert np.array_equal(back, src), nbytes
        d.free()
    libc = C.CDLL("libc.so.6")
    libc.mallopt(-3, 64 << 20)                      # M_MMAP_THRESHOLD: arrays of a few MB come from the brk heap, as in a long-lived process
    try:
        y = synth.plane(300, 500, synth.SEED0 + 9, "noise")
        want = oracle_lib.y_path(y)
        for it in range(40):
            ballast = [np.ones(rng.integers(1, 6) << 20, np.uint8) for _ in range(4)]      # grow the heap ...
            yy = y.copy()                                                                  # ... a heap-resident source
            got = S.y_upscale2x(yy)
            assert_bit_equal(got, want, "iteration %d" % it)
            del ballast, yy, got
            libc.malloc_trim(0)                                                            # ... and give its top back to the system
    finally:
        libc.mallopt(-3, 128 << 10)
