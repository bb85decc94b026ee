#!/usr/bin/env python3
"""A/B of the asynchronous ProcessSRCNN pair with page-locked buffers (3840x2160 RGB x2): blocking srcnn_process_u8 calls vs two
and three jobs in flight; run once per SRCNN_ASYNC_CHAIN setting (1 = host-resolved chain, the default; 2 = device-side stream
wait; 0 = unchained).  Results: profiles/r04_process_clock.txt.   SRCNN_ASYNC_CHAIN=1 python3 tools/async_ab.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, libsrcnn_amd as S
S.init(0); L = S.lib()
W, H = 3840, 2160
img = bench.synth_rgb(H, W, 0x5C0DE100)
pin = S.PinnedArray(img.shape); pin.array[...] = img
outs = [S.PinnedArray((2*H, 2*W, 3)) for _ in range(3)]
def blocking(n=16):
    for _ in range(3): S.check(L.srcnn_process_u8(pin.array.ctypes.data, W, H, 3, 2.0, 2, outs[0].array.ctypes.data, None))
    t0 = time.perf_counter()
    for _ in range(n): S.check(L.srcnn_process_u8(pin.array.ctypes.data, W, H, 3, 2.0, 2, outs[0].array.ctypes.data, None))
    return (time.perf_counter() - t0) * 1e3 / n
def asyncrun(depth, n=24):
    jobs = []
    t0 = time.perf_counter()
    for i in range(n):
        j = C.c_void_p()
        S.check(L.srcnn_process_u8_begin(pin.array.ctypes.data, W, H, 3, 2.0, 2, outs[i % 3].array.ctypes.data, None, C.byref(j)))
        jobs.append(j)
        if len(jobs) == depth: S.check(L.srcnn_process_u8_wait(jobs.pop(0)))
    while jobs: S.check(L.srcnn_process_u8_wait(jobs.pop(0)))
    return (time.perf_counter() - t0) * 1e3 / n
print("chain =", os.environ.get("SRCNN_ASYNC_CHAIN", "default"), "blocking %.2f" % blocking(), "async d2 %.2f %.2f" % (asyncrun(2), asyncrun(2)), "d3 %.2f" % asyncrun(3), flush=True)
