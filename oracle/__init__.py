"""CPU parity checker for the SRCNN Y path -- TEST INFRASTRUCTURE ONLY.

Import from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, nowhere else.
`Oracle()` binds oracle/_build/libsrcnn_oracle.so (the C restatement, srcnn_oracle.c);
`Reference()` binds oracle/_ref/libsrcnn_ref.so (the real reference compiled from
/root/reference/src by oracle/Makefile, present only if it was built in the dev container and
travelled with the snapshot).  Both expose the same stage-level calls on numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "_build", "libsrcnn_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libsrcnn_ref.so")
REF_SRC = "/root/reference/src"

FILTER_NEAREST, FILTER_BILINEAR, FILTER_BICUBIC, FILTER_LANCZOS3, FILTER_BSPLINE = range(5)

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(ref=True):
    """Compile the restatement (always) and the real reference (only where its tree exists)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir(REF_SRC):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def have_reference():
    return os.path.exists(REF_SO)


# ---- OpenMP team size ----------------------------------------------------------------------------
# Both libraries parallelise with `#pragma omp parallel for` and, left alone, start one thread per LOGICAL cpu of the host --
# 256 on the GPU boxes, whose containers may own far fewer cores.  For the small planes most tests check, a team that large
# spends its time spinning at barriers (a 130 x 130 case took ~1 s on the GPU box and 0.05 s on 8 cores): the team is
# therefore sized per call -- by the work, capped by the cpus this process may actually use and by 64 (layer 1 has 64
# parallel iterations, src/libsrcnn.cpp:791).  An explicit OMP_NUM_THREADS in the environment is left alone (bench.py's
# cpu_baseline sets it as SURVEY 8d asks).  Results do not depend on the team size (every reduction is per sample).
def cpu_budget():
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:                                         # cgroup v2 quota
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:                                     # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return max(1, n)


_gomp = None
_BUDGET = None


def set_team(pixels):
    """Size the OpenMP team of the calling thread for a call that produces `pixels` output samples."""
    global _gomp, _BUDGET
    if os.environ.get("OMP_NUM_THREADS"):
        return
    if _gomp is None:
        try:
            _gomp = C.CDLL("libgomp.so.1")
            _gomp.omp_set_num_threads.argtypes = [C.c_int]
        except OSError:
            _gomp = False
        _BUDGET = min(cpu_budget(), 64)
    if _gomp:
        _gomp.omp_set_num_threads(int(max(1, min(_BUDGET, pixels // 1024))))


class _Stages:
    """Stage-level calls shared by the restatement (prefix 'oracle_') and the reference ('ref_')."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        g = lambda n: getattr(self.lib, prefix + n)
        self._resample = g("resample")
        self._resample.argtypes = [_f32p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, _f32p, C.c_int]
        self._resample.restype = C.c_int
        for n in ("conv1", "conv2", "conv3"):
            f = g(n)
            f.argtypes = [_f32p, C.c_uint, C.c_uint, _f32p]
            f.restype = None
            setattr(self, "_" + n, f)
        self._y_path = g("y_path")
        self._y_path.argtypes = [_f32p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int,
                                 _f32p, C.c_void_p, C.c_void_p, C.c_void_p]
        self._y_path.restype = C.c_int
        self._weights = g("get_weights")
        self._weights.argtypes = [_f32p]
        self._weights.restype = None

    def weights(self):
        w = np.empty(8129, np.float32)
        self._weights(w)
        return w

    def resample(self, plane, dw, dh, filt=FILTER_BICUBIC):
        plane = np.ascontiguousarray(plane, np.float32)
        h, w = plane.shape
        out = np.zeros((dh, dw), np.float32)
        set_team(dw * dh)
        rc = self._resample(plane, w, h, dw, dh, out, filt)
        if rc != 0:
            raise RuntimeError("resample failed rc=%d" % rc)
        return out

    def conv1(self, y):
        y = np.ascontiguousarray(y, np.float32)
        h, w = y.shape
        out = np.empty((64, h, w), np.float32)
        set_team(w * h)
        self._conv1(y, w, h, out)
        return out

    def conv2(self, c1):
        c1 = np.ascontiguousarray(c1, np.float32)
        _, h, w = c1.shape
        out = np.empty((32, h, w), np.float32)
        set_team(w * h)
        self._conv2(c1, w, h, out)
        return out

    def conv3(self, c2):
        c2 = np.ascontiguousarray(c2, np.float32)
        _, h, w = c2.shape
        out = np.empty((h, w), np.float32)
        set_team(w * h)
        self._conv3(c2, w, h, out)
        return out

    def y_path(self, y, dw=None, dh=None, filt=FILTER_BICUBIC, taps=False):
        """Resample to (dw,dh) (default 2x) then the three convolutions.  taps=True also returns
        the upscaled plane and both activation stacks."""
        y = np.ascontiguousarray(y, np.float32)
        h, w = y.shape
        dw = 2 * w if dw is None else dw
        dh = 2 * h if dh is None else dh
        out = np.empty((dh, dw), np.float32)
        set_team(dw * dh)
        if taps:
            up = np.zeros((dh, dw), np.float32)
            c1 = np.empty((64, dh, dw), np.float32)
            c2 = np.empty((32, dh, dw), np.float32)
            rc = self._y_path(y, w, h, dw, dh, filt, out, up.ctypes.data, c1.ctypes.data, c2.ctypes.data)
        else:
            rc = self._y_path(y, w, h, dw, dh, filt, out, None, None, None)
        if rc != 0:
            raise RuntimeError("y_path failed rc=%d" % rc)
        return (out, up, c1, c2) if taps else out


class Oracle(_Stages):
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        super().__init__(ORACLE_SO, "oracle_")
        self.lib.oracle_dosrcnn.argtypes = [_u8p, C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_int,
                                            _u8p, C.c_void_p]
        self.lib.oracle_dosrcnn.restype = C.c_int
        self.lib.oracle_axis_window.argtypes = [C.c_int, C.c_uint, C.c_uint]
        self.lib.oracle_axis_window.restype = C.c_int
        self.lib.oracle_axis_table.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.oracle_axis_table.restype = None

    def process(self, rgb, mul=2.0, filt=FILTER_BICUBIC):
        """One doSRCNN pass on an interleaved u8 image (h,w,d).  Returns (rgb_out, conv_y_u8)."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w, d = rgb.shape
        mulf = np.float32(mul)
        dw, dh = int(np.float32(w) * mulf), int(np.float32(h) * mulf)
        out = np.empty((dh, dw, d), np.uint8)
        conv = np.empty((dh, dw), np.uint8)
        set_team(dw * dh)
        rc = self.lib.oracle_dosrcnn(rgb, w, h, d, float(mulf), filt, out, conv.ctypes.data)
        if rc != 0:
            raise RuntimeError("oracle_dosrcnn rc=%d" % rc)
        return out, conv

    def axis_table(self, dst_len, src_len, filt=FILTER_BICUBIC):
        win = self.lib.oracle_axis_window(filt, dst_len, src_len)
        left = np.zeros(dst_len, np.int32)
        right = np.zeros(dst_len, np.int32)
        w = np.zeros((dst_len, win + 1), np.float64)
        self.lib.oracle_axis_table(filt, dst_len, src_len, left.ctypes.data, right.ctypes.data, w.ctypes.data)
        return left, right, w


class Reference(_Stages):
    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " (build it in the dev container: make -C oracle ref)")
        super().__init__(REF_SO, "ref_")
        self.lib.ref_process.argtypes = [_u8p, C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_int, C.c_int,
                                         _u8p, C.c_uint, C.POINTER(C.c_uint),
                                         C.c_void_p, C.c_uint, C.POINTER(C.c_uint)]
        self.lib.ref_process.restype = C.c_int

    def process(self, rgb, mul=2.0, filt=FILTER_BICUBIC, step=False):
        """ConfigureFilterSRCNN + ProcessSRCNN of the real reference."""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        h, w, d = rgb.shape
        mulf = np.float32(mul)
        dw, dh = int(np.float32(w) * mulf), int(np.float32(h) * mulf)
        out = np.empty((dh, dw, d), np.uint8)
        conv = np.empty((dh, dw), np.uint8)
        osz, csz = C.c_uint(0), C.c_uint(0)
        set_team(dw * dh)
        rc = self.lib.ref_process(rgb, w, h, d, float(mulf), filt, int(step), out, out.size, C.byref(osz),
                                  conv.ctypes.data, conv.size, C.byref(csz))
        if rc != 0:
            raise RuntimeError("ProcessSRCNN rc=%d" % rc)
        assert osz.value == out.size and csz.value == conv.size, (osz.value, csz.value)
        return out, conv
