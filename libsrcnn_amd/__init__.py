"""libsrcnn_amd -- MI355X-native SRCNN Y-channel path behind the rageworx/libsrcnn interface.

The product is libsrcnn_amd/lib/libsrcnn_amd.so (hand-written gfx950 kernels + C ABI
include/srcnn_amd.h + the C++ drop-in symbols ProcessSRCNN / ConfigureFilterSRCNN).  This module
is only the ctypes binding used by tests/, bench.py and __graft_entry__.py; it mirrors the
reference's two public calls (src/libsrcnn.h:46-54) and exposes the planar-float Y entry points.

There is no CPU compute path here: if the shared object is missing, or no gfx950 device is
visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SRCNN_AMD_LIB: another build of the same library (A/B runs of two kernel versions on one box: tools/lib_ab.py)
LIB_PATH = os.environ.get("SRCNN_AMD_LIB") or os.path.join(_HERE, "lib", "libsrcnn_amd.so")

SRCNNF_Nearest, SRCNNF_Bilinear, SRCNNF_Bicubic, SRCNNF_Lanczos3, SRCNNF_Bspline = range(5)
MODE_STRICT, MODE_FAST, MODE_FAST_F16, MODE_RELAXED = 0, 1, 2, 3
RELAX_L1, RELAX_L2, RELAX_L3_X64, RELAX_L3_F32 = 1, 2, 4, 8

# The STABLE C ABI: every function include/srcnn_amd.h declares, frozen at ABI 5 (include/srcnn_amd.abi is the committed list;
# tests/test_abi.py holds header, list, this binding and the library's export table to each other)
STABLE_ABI_SYMBOLS = [
    "srcnn_abi_version", "srcnn_device_count", "srcnn_init", "srcnn_init_devices", "srcnn_context_count",
    "srcnn_context_device", "srcnn_set_context", "srcnn_get_context", "srcnn_shutdown", "srcnn_trim",
    "srcnn_last_error", "srcnn_set_mode", "srcnn_get_mode", "srcnn_device_name", "srcnn_set_workspace_limit",
    "srcnn_dev_alloc", "srcnn_dev_free", "srcnn_host_alloc_pinned", "srcnn_host_free_pinned", "srcnn_memcpy_h2d",
    "srcnn_memcpy_d2h", "srcnn_memset_dev", "srcnn_stream_create", "srcnn_stream_destroy", "srcnn_stream_sync",
    "srcnn_device_sync", "srcnn_event_create", "srcnn_event_destroy", "srcnn_event_record",
    "srcnn_stream_wait_event", "srcnn_event_elapsed_ms", "srcnn_profile_enable", "srcnn_profile_reset",
    "srcnn_profile_read", "srcnn_profile_read_context", "srcnn_y_upscale2x_f32_dev",
    "srcnn_y_upscale2x_f32_batch_dev", "srcnn_y_upscale2x_f32_band_dev", "srcnn_y_upscale2x_f32_node_dev",
    "srcnn_batch_graph_create", "srcnn_batch_graph_launch", "srcnn_batch_graph_destroy", "srcnn_y_path_f32_dev",
    "srcnn_resample_f32_dev", "srcnn_conv1_f32_dev", "srcnn_conv2_f32_dev", "srcnn_conv3_f32_dev",
    "srcnn_conv12_f32_dev", "srcnn_y_upscale2x_f32", "srcnn_y_upscale2x_f32_batch", "srcnn_y_upscale2x_f32_stream",
    "srcnn_y_path_f32", "srcnn_process_u8", "srcnn_process_u8_begin", "srcnn_process_u8_wait", "srcnn_delete_array",
    "srcnn_output_size", "srcnn_comm_unique_id", "srcnn_comm_init", "srcnn_comm_destroy", "srcnn_comm_rank",
    "srcnn_comm_gather_f32", "srcnn_comm_gatherv_f32", "srcnn_comm_gatherv_at_f32",
    "srcnn_comm_tiled_y_upscale2x_f32_dev", "srcnn_band_rows", "srcnn_tiled_piece", "srcnn_comm_allgather_f32",
    "srcnn_comm_barrier", "srcnn_comm_wait", "srcnn_comm_set_timeout_ms",
]
# instruments (include/srcnn_amd_debug.h): test hooks, diagnostics, the relaxation experiment -- no compatibility promise
DEBUG_SYMBOLS = [
    "srcnn_set_relaxation", "srcnn_axis_table", "srcnn_fused_diag", "srcnn_debug_counts", "srcnn_debug_settings",
    "srcnn_debug_clock_probe", "srcnn_debug_clock_read", "srcnn_debug_band_plan", "srcnn_debug_process_phases", "srcnn_debug_stream_mode",
]
C_ABI_SYMBOLS = STABLE_ABI_SYMBOLS + DEBUG_SYMBOLS          # everything the library exports besides the two C++ symbols
CXX_SYMBOLS = ["_Z20ConfigureFilterSRCNN15SRCNNFilterTypeb", "_Z12ProcessSRCNNPKhjjjfRPhRjPS1_Pj"]


class SrcnnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsrcnn_amd error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load the shared object (never builds, never falls back)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing -- run `python -m libsrcnn_amd.build` (hipcc, gfx950). "
                              "There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, u, f, sz, i = C.c_void_p, C.c_uint, C.c_float, C.c_size_t, C.c_int
        sig = {
            "srcnn_abi_version": (i, []), "srcnn_device_count": (i, []), "srcnn_init": (i, [i]),
            "srcnn_init_devices": (i, [C.POINTER(i), i]), "srcnn_context_count": (i, []), "srcnn_context_device": (i, [i]),
            "srcnn_set_context": (i, [i]), "srcnn_get_context": (i, []), "srcnn_trim": (i, []),
            "srcnn_y_upscale2x_f32_node_dev": (i, [vp, u, u, vp, i]),
            "srcnn_comm_gatherv_at_f32": (i, [vp, C.POINTER(sz), C.POINTER(sz), vp, i, vp]),
            "srcnn_comm_tiled_y_upscale2x_f32_dev": (i, [vp, u, u, vp, vp, i, i, vp]),
            "srcnn_band_rows": (i, [u, i, i, C.POINTER(u), C.POINTER(u)]),
            "srcnn_tiled_piece": (i, [u, u, i, i, i, i, C.POINTER(u), C.POINTER(u)]),
            "srcnn_debug_band_plan": (i, [u, u, u, i, C.POINTER(u), i]),
            "srcnn_shutdown": (None, []), "srcnn_last_error": (C.c_char_p, []), "srcnn_set_mode": (i, [i]),
            "srcnn_get_mode": (i, []), "srcnn_set_relaxation": (i, [u]), "srcnn_device_name": (i, [C.c_char_p, sz]),
            "srcnn_set_workspace_limit": (sz, [sz]),
            "srcnn_dev_alloc": (vp, [sz]), "srcnn_dev_free": (None, [vp]),
            "srcnn_host_alloc_pinned": (vp, [sz]), "srcnn_host_free_pinned": (None, [vp]),
            "srcnn_memcpy_h2d": (i, [vp, vp, sz, vp]), "srcnn_memcpy_d2h": (i, [vp, vp, sz, vp]),
            "srcnn_memset_dev": (i, [vp, i, sz, vp]),
            "srcnn_stream_create": (i, [C.POINTER(vp)]), "srcnn_stream_destroy": (i, [vp]),
            "srcnn_stream_sync": (i, [vp]), "srcnn_device_sync": (i, []),
            "srcnn_event_create": (i, [C.POINTER(vp)]), "srcnn_event_destroy": (i, [vp]),
            "srcnn_event_record": (i, [vp, vp]), "srcnn_event_elapsed_ms": (i, [vp, vp, C.POINTER(f)]),
            "srcnn_stream_wait_event": (i, [vp, vp]),
            "srcnn_profile_enable": (i, [i]), "srcnn_profile_reset": (i, []),
            "srcnn_profile_read": (i, [i, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)]),
            "srcnn_profile_read_context": (i, [i, i, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong)]),
            "srcnn_y_upscale2x_f32_dev": (i, [vp, u, u, vp, vp]),
            "srcnn_y_upscale2x_f32_batch_dev": (i, [vp, u, u, u, vp, vp]),
            "srcnn_y_upscale2x_f32_band_dev": (i, [vp, u, u, u, u, vp, vp]),
            "srcnn_batch_graph_create": (i, [vp, u, u, u, vp, vp, C.POINTER(vp)]),
            "srcnn_batch_graph_launch": (i, [vp]), "srcnn_batch_graph_destroy": (i, [vp]),
            "srcnn_y_path_f32_dev": (i, [vp, u, u, u, u, i, vp, vp]),
            "srcnn_resample_f32_dev": (i, [vp, u, u, u, u, i, vp, vp]),
            "srcnn_conv1_f32_dev": (i, [vp, u, u, vp, vp]), "srcnn_conv2_f32_dev": (i, [vp, u, u, vp, vp]),
            "srcnn_conv3_f32_dev": (i, [vp, u, u, vp, vp]), "srcnn_conv12_f32_dev": (i, [vp, u, u, vp, vp]),
            "srcnn_y_upscale2x_f32": (i, [vp, u, u, vp]), "srcnn_y_upscale2x_f32_batch": (i, [vp, u, u, u, vp]),
            "srcnn_y_path_f32": (i, [vp, u, u, u, u, i, vp]),
            "srcnn_y_upscale2x_f32_stream": (i, [vp, u, u, u, vp, i]),
            "srcnn_process_u8": (i, [vp, u, u, u, f, i, vp, vp]),
            "srcnn_process_u8_begin": (i, [vp, u, u, u, f, i, vp, vp, C.POINTER(vp)]), "srcnn_process_u8_wait": (i, [vp]),
            "srcnn_delete_array": (None, [vp]),
            "srcnn_output_size": (i, [u, u, f, i, C.POINTER(u), C.POINTER(u)]),
            "srcnn_axis_table": (i, [i, u, u, vp, vp, vp]),
            "srcnn_comm_unique_id": (i, [vp]), "srcnn_comm_init": (i, [vp, i, i]), "srcnn_comm_destroy": (i, []),
            "srcnn_comm_gather_f32": (i, [vp, sz, vp, i, vp]), "srcnn_comm_allgather_f32": (i, [vp, sz, vp, vp]),
            "srcnn_comm_gatherv_f32": (i, [vp, C.POINTER(sz), vp, i, vp]),
            "srcnn_comm_rank": (i, [C.POINTER(i), C.POINTER(i)]),
            "srcnn_debug_counts": (i, [C.POINTER(i), C.POINTER(i)]),
            "srcnn_fused_diag": (i, [vp, u, u, vp, vp, vp]),
            "srcnn_debug_clock_probe": (i, [i]), "srcnn_debug_clock_read": (i, [i, vp, vp, i]),
            "srcnn_debug_settings": (i, [C.c_char_p, sz, i]),
            "srcnn_debug_process_phases": (i, [C.POINTER(C.c_double), i]),
            "srcnn_debug_stream_mode": (i, [C.POINTER(u), C.POINTER(u), C.POINTER(i)]),
            "srcnn_comm_barrier": (i, [vp]), "srcnn_comm_wait": (i, [vp]), "srcnn_comm_set_timeout_ms": (i, [i]),
        }
        for name, (res, args) in sig.items():
            if name in DEBUG_SYMBOLS and not hasattr(L, name) and os.environ.get("SRCNN_AMD_LIB"):
                continue                  # an older build loaded for an A/B run (tools/lib_ab.py): it may lack newer instruments
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        cfg = getattr(L, CXX_SYMBOLS[0])
        cfg.restype, cfg.argtypes = None, [i, C.c_bool]
        prc = getattr(L, CXX_SYMBOLS[1])
        # references are passed as pointers in the Itanium C++ ABI
        prc.restype = i
        prc.argtypes = [vp, u, u, u, f, C.POINTER(vp), C.POINTER(u), C.POINTER(vp), C.POINTER(u)]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise SrcnnError(rc, lib().srcnn_last_error().decode("utf-8", "replace"))


def init(device=0):
    check(lib().srcnn_init(int(device)))


def init_devices(devices=None):
    """One context per entry of `devices` (HIP device ids; an id may repeat = virtual contexts); None = every visible device."""
    if devices is None:
        check(lib().srcnn_init_devices(None, 0))
    else:
        arr = (C.c_int * len(devices))(*devices)
        check(lib().srcnn_init_devices(arr, len(devices)))
    return lib().srcnn_context_count()


def context_count():
    return lib().srcnn_context_count()


def set_context(k):
    prev = lib().srcnn_set_context(int(k))
    if prev < 0:
        raise SrcnnError(prev, lib().srcnn_last_error().decode())
    return prev


def shutdown():
    lib().srcnn_shutdown()


def device_count():
    return lib().srcnn_device_count()


def device_name():
    buf = C.create_string_buffer(256)
    check(lib().srcnn_device_name(buf, 256))
    return buf.value.decode()


def set_mode(mode):
    prev = lib().srcnn_set_mode(int(mode))
    if prev < 0:
        raise SrcnnError(prev, lib().srcnn_last_error().decode())
    return prev


def set_relaxation(mask):
    """Which roundings MODE_RELAXED gives up (RELAX_* bits); returns the previous mask."""
    prev = lib().srcnn_set_relaxation(int(mask))
    if prev < 0:
        raise SrcnnError(prev, lib().srcnn_last_error().decode())
    return prev


def sync():
    check(lib().srcnn_device_sync())


STAGES = ("resample", "conv12", "conv3")


def profile_enable(on=True):
    return lib().srcnn_profile_enable(1 if on else 0)


def profile_reset():
    check(lib().srcnn_profile_reset())


def clock_probe(on=True):
    return lib().srcnn_debug_clock_probe(1 if on else 0)


def clock_read(context=0, cap=8192):
    """[(MHz, microseconds)] of every layer-1+2 launch since clock_probe(True), in launch order."""
    cyc = np.zeros(cap, np.uint64); tk = np.zeros(cap, np.uint64)
    n = lib().srcnn_debug_clock_read(int(context), cyc.ctypes.data, tk.ctypes.data, cap)
    if n < 0:
        raise SrcnnError(n, lib().srcnn_last_error().decode())
    n = min(n, cap)
    t = np.maximum(tk[:n].astype(np.float64), 1.0)
    return [(float(c) / float(x) * 100.0, float(x) / 100.0) for c, x in zip(cyc[:n].astype(np.float64), t)]


def profile_read_context(k):
    """The same for context k alone (which device of a node-level call is the straggler)."""
    out = {}
    for st, name in enumerate(STAGES):
        ms, n = C.c_double(), C.c_ulonglong()
        check(lib().srcnn_profile_read_context(int(k), st, C.byref(ms), C.byref(n)))
        out[name] = (ms.value, n.value)
    return out


def profile_read():
    """{stage: (total_ms, launches)} accumulated by HIP events on the launch stream since the last reset."""
    out = {}
    for k, name in enumerate(STAGES):
        ms, n = C.c_double(0), C.c_ulonglong(0)
        check(lib().srcnn_profile_read(k, C.byref(ms), C.byref(n)))
        out[name] = (ms.value, n.value)
    return out


# ------------------------------------------------------------------------------------------------
# device buffers / streams / events: thin RAII over the C ABI (no torch needed)
# ------------------------------------------------------------------------------------------------
class DeviceBuffer:
    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib().srcnn_dev_alloc(self.nbytes)
        if not self.ptr:
            raise SrcnnError(-202, lib().srcnn_last_error().decode())

    @classmethod
    def from_numpy(cls, arr):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        check(lib().srcnn_memcpy_h2d(b.ptr, arr.ctypes.data, arr.nbytes, None))
        return b

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        check(lib().srcnn_memcpy_h2d(self.ptr + offset, arr.ctypes.data, arr.nbytes, None))

    def to_numpy(self, dtype, shape, offset=0):
        out = np.empty(shape, dtype)
        assert offset + out.nbytes <= self.nbytes
        check(lib().srcnn_memcpy_d2h(out.ctypes.data, self.ptr + offset, out.nbytes, None))
        return out

    def free(self):
        if self.ptr:
            lib().srcnn_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Stream:
    def __init__(self):
        h = C.c_void_p()
        check(lib().srcnn_stream_create(C.byref(h)))
        self.handle = h.value

    def sync(self):
        check(lib().srcnn_stream_sync(self.handle))

    def destroy(self):
        if self.handle:
            check(lib().srcnn_stream_destroy(self.handle))
            self.handle = None


class Event:
    def __init__(self):
        h = C.c_void_p()
        check(lib().srcnn_event_create(C.byref(h)))
        self.handle = h.value

    def record(self, stream=None):
        check(lib().srcnn_event_record(self.handle, stream.handle if stream else None))

    def elapsed_ms(self, stop):
        ms = C.c_float()
        check(lib().srcnn_event_elapsed_ms(self.handle, stop.handle, C.byref(ms)))
        return ms.value


# ------------------------------------------------------------------------------------------------
# numpy-level helpers over the C ABI (host arrays in, host arrays out)
# ------------------------------------------------------------------------------------------------
def _plane(a):
    a = np.ascontiguousarray(a, np.float32)
    assert a.ndim == 2, a.shape
    return a


def y_upscale2x(y):
    """srcnn_y_upscale2x_f32: 2x Mitchell upscale + the three convolutions of one float Y plane."""
    y = _plane(y)
    h, w = y.shape
    out = np.empty((2 * h, 2 * w), np.float32)
    check(lib().srcnn_y_upscale2x_f32(y.ctypes.data, w, h, out.ctypes.data))
    return out


def y_upscale2x_batch(frames):
    frames = np.ascontiguousarray(frames, np.float32)
    n, h, w = frames.shape
    out = np.empty((n, 2 * h, 2 * w), np.float32)
    check(lib().srcnn_y_upscale2x_f32_batch(frames.ctypes.data, w, h, n, out.ctypes.data))
    return out


def y_upscale2x_stream(frames, use_graph=True):
    """srcnn_y_upscale2x_f32_stream.  use_graph: False / 0 = plain launches, True / 2 = hipGraph replay whatever it costs,
    "auto" / 1 = replay kept only while it is cheap in host CPU (stream_mode() says what ran)."""
    frames = np.ascontiguousarray(frames, np.float32)
    n, h, w = frames.shape
    out = np.empty((n, 2 * h, 2 * w), np.float32)
    g = 1 if use_graph == "auto" else (int(use_graph) if isinstance(use_graph, int) and not isinstance(use_graph, bool) else (2 if use_graph else 0))
    check(lib().srcnn_y_upscale2x_f32_stream(frames.ctypes.data, w, h, n, out.ctypes.data, g))
    return out


def stream_mode():
    """(frames replayed from a hipGraph, frames launched plainly, fell back?) of the process's last stream call."""
    g, p, f = C.c_uint(0), C.c_uint(0), C.c_int(0)
    check(lib().srcnn_debug_stream_mode(C.byref(g), C.byref(p), C.byref(f)))
    return g.value, p.value, bool(f.value)


def y_path(y, dw, dh, filt=SRCNNF_Bicubic):
    y = _plane(y)
    h, w = y.shape
    out = np.empty((dh, dw), np.float32)
    check(lib().srcnn_y_path_f32(y.ctypes.data, w, h, dw, dh, filt, out.ctypes.data))
    return out


def y_upscale2x_band(y, row0, rows):
    y = _plane(y)
    h, w = y.shape
    din = DeviceBuffer.from_numpy(y)
    dout = DeviceBuffer(rows * 2 * w * 4)
    check(lib().srcnn_y_upscale2x_f32_band_dev(din.ptr, w, h, row0, rows, dout.ptr, None))
    sync()
    return dout.to_numpy(np.float32, (rows, 2 * w))


def _stage(fn, src, out_shape, *dims):
    src = np.ascontiguousarray(src, np.float32)
    din = DeviceBuffer.from_numpy(src)
    dout = DeviceBuffer(int(np.prod(out_shape)) * 4)
    check(fn(din.ptr, *dims, dout.ptr, None))
    sync()
    return dout.to_numpy(np.float32, out_shape)


def resample(y, dw, dh, filt=SRCNNF_Bicubic):
    y = _plane(y)
    h, w = y.shape
    src = DeviceBuffer.from_numpy(y)
    dst = DeviceBuffer(dw * dh * 4)
    check(lib().srcnn_resample_f32_dev(src.ptr, w, h, dw, dh, filt, dst.ptr, None))
    sync()
    return dst.to_numpy(np.float32, (dh, dw))


def conv1(y):
    y = _plane(y)
    h, w = y.shape
    return _stage(lib().srcnn_conv1_f32_dev, y, (64, h, w), w, h)


def conv2(c1):
    _, h, w = c1.shape
    return _stage(lib().srcnn_conv2_f32_dev, c1, (32, h, w), w, h)


def conv3(c2):
    _, h, w = c2.shape
    return _stage(lib().srcnn_conv3_f32_dev, c2, (h, w), w, h)


def conv12(y):
    y = _plane(y)
    h, w = y.shape
    return _stage(lib().srcnn_conv12_f32_dev, y, (32, h, w), w, h)


def axis_table(dst_len, src_len, filt=SRCNNF_Bicubic):
    win = lib().srcnn_axis_table(filt, dst_len, src_len, None, None, None)
    if win < 0:
        raise SrcnnError(win, lib().srcnn_last_error().decode())
    left = np.zeros(dst_len, np.int32)
    right = np.zeros(dst_len, np.int32)
    w = np.zeros((dst_len, win + 1), np.float64)
    lib().srcnn_axis_table(filt, dst_len, src_len, left.ctypes.data, right.ctypes.data, w.ctypes.data)
    return left, right, w


def output_size(w, h, multiply, stepscale=False):
    ow, oh = C.c_uint(0), C.c_uint(0)
    rc = lib().srcnn_output_size(w, h, float(np.float32(multiply)), 1 if stepscale else 0, C.byref(ow), C.byref(oh))
    if rc != 0:
        raise SrcnnError(rc, "srcnn_output_size")
    return ow.value, oh.value


def process_u8(rgb, multiply=2.0, filt=SRCNNF_Bicubic, want_conv=True):
    """srcnn_process_u8: one doSRCNN pass, interleaved u8 (h,w,d) in, (rgb_out, conv_y|None) out."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, d = rgb.shape
    m = np.float32(multiply)
    dw, dh = int(np.float32(w) * m), int(np.float32(h) * m)
    out = np.empty((dh, dw, d), np.uint8)
    conv = np.empty((dh, dw), np.uint8) if want_conv else None
    check(lib().srcnn_process_u8(rgb.ctypes.data, w, h, d, float(m), filt, out.ctypes.data,
                                 conv.ctypes.data if want_conv else None))
    return out, conv


class PinnedArray:
    """A numpy array on page-locked host memory from srcnn_host_alloc_pinned: srcnn_process_u8 / ProcessJob move such buffers
    to and from the device without staging copies.  Keep the object alive while the array is in use; free() when done."""

    def __init__(self, shape, dtype=np.uint8):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.ptr = lib().srcnn_host_alloc_pinned(max(1, self.nbytes))
        if not self.ptr:
            raise SrcnnError(-202, lib().srcnn_last_error().decode())
        raw = (C.c_ubyte * max(1, self.nbytes)).from_address(self.ptr)
        self.array = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib().srcnn_host_free_pinned(self.ptr)
            self.ptr = None


class ProcessJob:
    """srcnn_process_u8_begin / _wait: the image is being produced; result() blocks and returns (rgb_out, conv_y|None).

    The object owns the buffers a native worker thread reads and writes, so it must never be reclaimed while that thread is
    still running: result(), leaving a `with` block and the finaliser all end in srcnn_process_u8_wait (the finaliser is what
    covers a job dropped without result() -- e.g. the second constructor of `[ProcessJob(a), ProcessJob(b)]` raising)."""

    def __init__(self, rgb, multiply=2.0, filt=SRCNNF_Bicubic, want_conv=True, out=None, conv=None):
        self.job = None
        self.rgb = np.ascontiguousarray(rgb, np.uint8)          # kept alive until the job is done
        h, w, d = self.rgb.shape
        m = np.float32(multiply)
        dw, dh = int(np.float32(w) * m), int(np.float32(h) * m)
        self.out = out if out is not None else np.empty((dh, dw, d), np.uint8)
        self.conv = (conv if conv is not None else np.empty((dh, dw), np.uint8)) if want_conv else None
        job = C.c_void_p()
        check(lib().srcnn_process_u8_begin(self.rgb.ctypes.data, w, h, d, float(m), filt, self.out.ctypes.data,
                                           self.conv.ctypes.data if want_conv else None, C.byref(job)))
        self.job = job

    def _wait(self):
        """Join the native job (idempotent); returns its code."""
        if self.job is None:
            return 0
        job, self.job = self.job, None
        return lib().srcnn_process_u8_wait(job)

    def result(self):
        check(self._wait())
        return self.out, self.conv

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._wait()
        return False

    def __del__(self):
        try:
            self._wait()          # the worker thread must be done with rgb / out / conv before they are freed
        except Exception:
            pass


def debug_settings(markdown=False):
    """The SRCNN_* switches this process runs with (srcnn_debug_settings): text, one line per switch.  No device needed."""
    n = lib().srcnn_debug_settings(None, 0, 1 if markdown else 0)
    buf = C.create_string_buffer(n + 1)
    lib().srcnn_debug_settings(buf, n + 1, 1 if markdown else 0)
    return buf.value.decode()


# ------------------------------------------------------------------------------------------------
# The reference's public API, same names and argument meaning (src/libsrcnn.h:46-54), called through
# the exported C++ symbols.
# ------------------------------------------------------------------------------------------------
def ConfigureFilterSRCNN(ftype, stepscale=False):
    getattr(lib(), CXX_SYMBOLS[0])(int(ftype), bool(stepscale))


def ProcessSRCNN(refbuff, w, h, d, multiply, want_conv=True):
    """Returns (retcode, outbuff ndarray|None, convbuff ndarray|None) -- the reference's out-params."""
    L = lib()
    if refbuff is not None:
        refbuff = np.ascontiguousarray(refbuff, np.uint8)
    out, outsz = C.c_void_p(), C.c_uint(0)
    conv, convsz = C.c_void_p(), C.c_uint(0)
    rc = getattr(L, CXX_SYMBOLS[1])(refbuff.ctypes.data if refbuff is not None else None, w, h, d,
                                    float(np.float32(multiply)), C.byref(out), C.byref(outsz),
                                    C.byref(conv) if want_conv else None, C.byref(convsz) if want_conv else None)
    o = c = None
    if rc == 0:
        o = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_ubyte)), (outsz.value,)).copy()
        L.srcnn_delete_array(out)
        if want_conv and conv.value:
            c = np.ctypeslib.as_array(C.cast(conv, C.POINTER(C.c_ubyte)), (convsz.value,)).copy()
            L.srcnn_delete_array(conv)
    return rc, o, c
