#!/usr/bin/env python3
"""A/B of two BUILDS of the library on one box: tools/lib_ab.py a.so b.so [rounds]   (alternating child processes).

Each child times the layer-1+2 kernel alone and the whole strict path on the headline frame (3840x2160 -> 7680x4320),
with HIP events, and prints a checksum of the result so that an A/B between builds that disagree is visible at once."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, hashlib
sys.path.insert(0, %r)
import numpy as np, libsrcnn_amd as S
from libsrcnn_amd import synth
S.init(0); L = S.lib()
h, w = 2160, 3840; H, W = 2*h, 2*w
din = S.DeviceBuffer.from_numpy(synth.plane(h, w, synth.SEED0, "smooth"))
up = S.DeviceBuffer(H*W*4); c2 = S.DeviceBuffer(32*H*W*4); out = S.DeviceBuffer(H*W*4)
S.check(L.srcnn_resample_f32_dev(din.ptr, w, h, W, H, 2, up.ptr, None))
def t(fn, reps=4, n=4):
    fn(); S.sync(); best = 1e9
    for _ in range(reps):
        e0, e1 = S.Event(), S.Event(); e0.record()
        for _ in range(n): fn()
        e1.record(); best = min(best, e0.elapsed_ms(e1)/n)
    return best
c12 = t(lambda: S.check(L.srcnn_conv12_f32_dev(up.ptr, W, H, c2.ptr, None)))
c3 = t(lambda: S.check(L.srcnn_conv3_f32_dev(c2.ptr, W, H, out.ptr, None)))
whole = t(lambda: S.check(L.srcnn_y_upscale2x_f32_dev(din.ptr, w, h, out.ptr, None)))
sha = hashlib.sha256(out.to_numpy(np.float32, (H, W)).tobytes()).hexdigest()[:12]
print("%%-28s conv12 %%.3f ms  conv3 %%.3f ms  whole %%.3f ms = %%.0f MPix/s  sha %%s" %% (os.path.basename(os.environ["SRCNN_AMD_LIB"]), c12, c3, whole, H*W/1e3/whole, sha), flush=True)
''' % ROOT
libs = [os.path.abspath(a) for a in sys.argv[1:3]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for _ in range(rounds):
    for lib in libs:
        subprocess.call([sys.executable, "-c", code], env=dict(os.environ, SRCNN_AMD_LIB=lib))
