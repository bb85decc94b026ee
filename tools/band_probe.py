#!/usr/bin/env python3
"""Why do the layer kernels take 20 % longer inside ProcessSRCNN than in the resident frame stream?  Separates banding from
idle gaps: whole frames vs 4 bands per frame, back to back vs with host-side pauses between frames (the library's own
per-stage HIP-event timers, srcnn_profile_*)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libsrcnn_amd as S
from libsrcnn_amd import synth

S.init(0)
L = S.lib()
w, h = 3840, 2160
y = synth.plane(h, w, synth.SEED0, "smooth")
d_in = S.DeviceBuffer.from_numpy(y)
d_out = S.DeviceBuffer(4 * w * h * 4)
cuts = [0, 1728, 3024, 3792, 4320]


def whole():
    S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_out.ptr, None))


def bands():
    for a, b in zip(cuts[:-1], cuts[1:]):
        S.check(L.srcnn_y_upscale2x_f32_band_dev(d_in.ptr, w, h, a, b - a, d_out.ptr + a * 2 * w * 4, None))


def run(fn, pause, frames=10):
    for _ in range(2):
        fn()
    S.sync()
    S.profile_reset(); S.profile_enable(True)
    for _ in range(frames):
        fn()
        if pause:
            S.sync()
            time.sleep(pause)
    S.sync()
    S.profile_enable(False)
    p = S.profile_read()
    return {k: round(v[0] / frames, 3) for k, v in p.items()}


res = {}
for name, fn in (("whole", whole), ("4bands", bands)):
    for pause in (0, 0.005, 0.02, 0.1):
        res["%s pause %g s" % (name, pause)] = run(fn, pause)
print(json.dumps(res, indent=1))
