#!/usr/bin/env python3
"""Where the fused strict kernel's time goes: FS_PHASES=1 (layers 1+2 into the ring only), =2 (layer 3 from a stale ring only), =3."""
import ctypes as C, os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE)); sys.path.insert(0, ROOT)
import libsrcnn_amd as S
from libsrcnn_amd import synth
S.init(0); L = S.lib()
F = C.CDLL(os.path.join(HERE, "libfused_strict.so"))
F.fused_strict_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
w = np.fromfile(os.path.join(ROOT, "tests", "golden", "weights_f32.bin"), dtype="<f4")
assert F.fused_strict_init(w.ctypes.data_as(C.c_void_p)) == 0
h, w_ = 2160, 3840; H, W = 2 * h, 2 * w_
dup = S.DeviceBuffer.from_numpy(synth.plane(H, W, synth.SEED0, "smooth")); do = S.DeviceBuffer(H * W * 4)
def run(): assert F.fused_strict_run(dup.ptr, W, H, do.ptr, 4, None) == 0
run(); S.sync(); best = 1e9
for _ in range(3):
    e0, e1 = S.Event(), S.Event(); e0.record()
    for _ in range(4): run()
    e1.record(); best = min(best, e0.elapsed_ms(e1) / 4)
print("FS_PHASES=%s: %.3f ms" % (os.environ.get("FS_PHASES", "3"), best))
