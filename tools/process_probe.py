#!/usr/bin/env python3
"""ProcessSRCNN on one synthetic RGB(A) image, a few times: the command the colour-shell profiles
(profiles/rNN_process_*) are collected from.

    python3 tools/process_probe.py [--size 3840x2160] [--depth 3] [--scale 2.0] [--reps 6] [--conv]
Prints one JSON line with the wall time of every call (the reference's only timing facility is exactly this wall
time, src/test.cpp:653-672) and the process CPU time per call.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--scale", type=float, default=2.0)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--conv", action="store_true", help="also ask for the conv-Y plane")
    args = ap.parse_args()
    w, h = (int(v) for v in args.size.split("x"))
    import bench
    import libsrcnn_amd as S
    S.init(0)
    L = S.lib()
    img = bench.synth_rgb(h, w, 0x5C0DE000 + h)
    if args.depth == 4:
        img = np.concatenate([img, np.full((h, w, 1), 200, np.uint8)], axis=2)
    S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
    fn = getattr(L, S.CXX_SYMBOLS[1])
    ts, cs = [], []
    for _ in range(args.reps):
        o, osz, cv, csz = C.c_void_p(), C.c_uint(0), C.c_void_p(), C.c_uint(0)
        c0 = time.process_time(); t0 = time.perf_counter()
        rc = fn(img.ctypes.data, w, h, args.depth, args.scale, C.byref(o), C.byref(osz),
                C.byref(cv) if args.conv else None, C.byref(csz) if args.conv else None)
        ts.append(time.perf_counter() - t0); cs.append(time.process_time() - c0)
        assert rc == 0, (rc, L.srcnn_last_error())
        L.srcnn_delete_array(o)
        if args.conv:
            L.srcnn_delete_array(cv)
    print(json.dumps({"image": [w, h, args.depth], "scale": args.scale, "wall_ms": [round(t * 1e3, 3) for t in ts],
                      "cpu_ms": [round(t * 1e3, 3) for t in cs], "best_ms": round(min(ts[1:] or ts) * 1e3, 3)}))


if __name__ == "__main__":
    main()
