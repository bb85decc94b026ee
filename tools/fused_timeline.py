#!/usr/bin/env python3
"""Per-row phase timeline of the fused fp16 kernel (workgroup 0), from in-kernel s_memtime stamps of a DIAG build:
how long layer 1, layers 2+3 and the gather take per row for each wave, and how the two waves that share a SIMD
(w and w+4) overlap in time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import libsrcnn_amd as S
from libsrcnn_amd import synth

S.init(0)
h, w = 4320, 7680
up = synth.plane(h, w, 5, "smooth")
d_up = S.DeviceBuffer.from_numpy(up)
d_out = S.DeviceBuffer(up.nbytes)
d_dbg = S.DeviceBuffer(8 * 64 * 4 * 8)
L = S.lib()
for it in range(3):
    S.check(L.srcnn_memset_dev(d_dbg.ptr, 0, 8 * 64 * 4 * 8, None))
    S.check(L.srcnn_fused_diag(d_up.ptr, w, h, d_out.ptr, d_dbg.ptr, None))
    S.sync()
t = d_dbg.to_numpy(np.uint64, (8, 64, 4)).astype(np.int64)
for wv in range(8):
    row = np.diff(t[wv, 8:56, 0])
    l1 = t[wv, 8:55, 1] - t[wv, 8:55, 0]
    l23 = t[wv, 9:56, 0] - t[wv, 8:55, 1]
    print("wave %d: row period median %6.0f  min %6.0f  max %6.0f cycles | layer 1 %6.0f  layers 2+3+gather %6.0f | start offset vs wave 0 at row 8: %7d" %
          (wv, np.median(row), row.min(), row.max(), np.median(l1), np.median(l23), int(t[wv, 8, 0] - t[0, 8, 0])))
