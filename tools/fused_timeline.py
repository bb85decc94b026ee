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
t0 = t[:, :, 0].min()
rows = slice(8, 56)
for wv in range(8):
    a, b, c, d = (t[wv, rows, k] for k in range(4))
    row = np.diff(t[wv, rows, 0])
    print("wave %d: row period %6.0f  | layer1 %6.0f  layers2+3 %6.0f  gather+store %5.0f   (cycles, median over rows)" %
          (wv, np.median(row), np.median(b - a), np.median(c - b), np.median(d - c)))
print("start offsets of rows 8..12 relative to wave 0 (cycles): partner waves are (0,4) (1,5) (2,6) (3,7)")
for wv in range(8):
    print("  wave %d:" % wv, [int(x) for x in (t[wv, 8:13, 0] - t[0, 8:13, 0])])
# overlap of layer-1 phases of partner waves 0 and 4
def intervals(wv, k0, k1):
    return [(int(t[wv, r, k0]), int(t[wv, r, k1])) for r in range(8, 56)]
def overlap(A, B):
    tot = 0
    for a0, a1 in A:
        for b0, b1 in B:
            tot += max(0, min(a1, b1) - max(a0, b0))
    return tot
for p in range(4):
    A, B = intervals(p, 0, 1), intervals(p + 4, 0, 1)
    la = sum(x1 - x0 for x0, x1 in A)
    print("waves %d/%d: layer-1 time %d cycles, of which %d (%.0f%%) coincide with the partner's layer 1" %
          (p, p + 4, la, overlap(A, B), 100.0 * overlap(A, B) / la))
