#!/usr/bin/env python3
"""Randomized check of the single-process tiled frame: srcnn_y_upscale2x_f32_node_dev over K (virtual) contexts, random plane
sizes, sub-band counts and root contexts, against the whole-frame call of the same plane, bit for bit.
    python tools/node_tiled_campaign.py [cases] [contexts] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import libsrcnn_amd as S
from libsrcnn_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
ndev = S.lib().srcnn_device_count()
S.init_devices([i % max(ndev, 1) for i in range(k)])
L = S.lib()
rng = np.random.default_rng(seed)
bad = 0
for case in range(n):
    big = rng.random() < 0.2
    h = int(rng.integers(200, 900)) if big else int(rng.integers(1, 160))
    w = int(rng.integers(300, 1600)) if big else int(rng.integers(1, 260))
    nsub = int(rng.integers(1, 17)); root = int(rng.integers(0, k))
    y = synth.plane(h, w, int(rng.integers(0, 1 << 30)), "noise" if rng.random() < 0.5 else "smooth")
    S.set_context(root)
    d_in = S.DeviceBuffer.from_numpy(y); d_ref = S.DeviceBuffer(4 * h * w * 4); d_out = S.DeviceBuffer(4 * h * w * 4)
    S.check(L.srcnn_memset_dev(d_out.ptr, 0xFF, 4 * h * w * 4, None))
    S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_ref.ptr, None)); S.sync()
    S.check(L.srcnn_y_upscale2x_f32_node_dev(d_in.ptr, w, h, d_out.ptr, nsub))
    a = d_out.to_numpy(np.float32, (2 * h, 2 * w)); b = d_ref.to_numpy(np.float32, (2 * h, 2 * w))
    if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
        bad += 1
        print("MISMATCH case %d: %dx%d nsub %d root %d" % (case, w, h, nsub, root), flush=True)
    d_in.free(); d_ref.free(); d_out.free()
    if case % 20 == 19:
        print("  ... %d cases" % (case + 1), flush=True)
print("node-tiled campaign: %d cases over %d contexts, %d mismatches" % (n, k, bad))
sys.exit(1 if bad else 0)
