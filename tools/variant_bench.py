#!/usr/bin/env python3
"""A/B the layer-1+2 kernel variants in one process per variant (env var is read at init)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np, libsrcnn_amd as S
from libsrcnn_amd import synth
S.init(0); L = S.lib()
h, w = 2160, 3840; H, W = 2*h, 2*w
kind = os.environ.get("SRCNN_BENCH_KIND", "smooth")
up = S.DeviceBuffer.from_numpy(synth.plane(H, W, synth.SEED0, kind))
c2 = S.DeviceBuffer(32*H*W*4)
def run(): S.check(L.srcnn_conv12_f32_dev(up.ptr, W, H, c2.ptr, None))
run(); S.sync()
ts = []
for rep in range(3):
    e0, e1 = S.Event(), S.Event(); e0.record()
    for _ in range(3): run()
    e1.record(); ts.append(e0.elapsed_ms(e1)/3)
chk = c2.to_numpy(np.float32, (32, 8, W))   # first 8 rows of every plane
import hashlib
print("variant %%s (%%s): conv12 %%.3f ms (min of 3x3)  sha %%s" %% (os.environ.get("SRCNN_CONV12_VARIANT","0"), kind, min(ts), hashlib.sha256(chk.tobytes()).hexdigest()[:12]))
''' % ROOT
for v in sys.argv[1:] or ["0", "1", "2", "3", "4"]:
    for kind in ("smooth", "noise"):
        env = dict(os.environ, SRCNN_CONV12_VARIANT=v, SRCNN_BENCH_KIND=kind)
        subprocess.call([sys.executable, "-c", code], env=env)
