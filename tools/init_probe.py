#!/usr/bin/env python3
"""Where does the start of a process go?  HIP runtime init (hipGetDeviceCount), loading the library, srcnn_init (its first touch
of a device symbol loads the 1 MB code object: 50-250 ms by box and page-cache state; SRCNN_TRACE=1 prints its phases), the first
tiny call, the second.  Round 6: page-locking the 32 MB of bounce slots used to sit in there as well (5-80 ms): they now grow
with the largest copy seen, and the 50 KB weight image of srcnn_init takes 2 MB.

    python3 tools/init_probe.py            (on the GPU box; SRCNN_AMD_LIB=... for another build)"""
import ctypes as C, time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
t0 = time.perf_counter()
hip = C.CDLL("libamdhip64.so.7")
t1 = time.perf_counter()
n = C.c_int(0); hip.hipGetDeviceCount(C.byref(n))
t2 = time.perf_counter()
hip.hipSetDevice(0); hip.hipFree(None)
t3 = time.perf_counter()
import libsrcnn_amd as S
L = S.lib()
t4 = time.perf_counter()
S.init(0)
t5 = time.perf_counter()
import numpy as np
y = np.zeros((64, 64), np.float32)
S.y_upscale2x(y)
t6 = time.perf_counter()
S.y_upscale2x(y)
t7 = time.perf_counter()
print("dlopen hip %.1f ms | hipGetDeviceCount %.1f | hipSetDevice+hipFree(0) %.1f | load libsrcnn_amd (+numpy import) %.1f | srcnn_init %.1f | first tiny call %.1f | second %.2f"
      % tuple(1e3 * (b - a) for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6), (t6, t7))))
