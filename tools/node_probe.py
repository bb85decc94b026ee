#!/usr/bin/env python3
"""Node-level calls from ONE process.  On a one-GPU box the K contexts alias device 0, so this measures the cost of the
dealing (threads, lanes, peer copies onto the same device), not a speed-up; on an 8-GPU node run it with --devices all.

    python3 tools/node_probe.py [--devices all | 0,0,0,0 | ...] [--reps 5]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import libsrcnn_amd as S
from libsrcnn_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="0")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    n = S.init_devices(None if args.devices == "all" else [int(v) for v in args.devices.split(",")])
    L = S.lib()
    out = {"contexts": n, "devices": [L.srcnn_context_device(k) for k in range(n)]}
    # (1) ProcessSRCNN, 4K RGB x2
    img = bench.synth_rgb(2160, 3840, 0x5C0DE000 + 2160)
    S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
    fn = getattr(L, S.CXX_SYMBOLS[1])
    ts, cs = [], []
    S.profile_reset(); S.profile_enable(True)
    for _ in range(args.reps + 1):
        o, osz = C.c_void_p(), C.c_uint(0)
        c0 = time.process_time(); t0 = time.perf_counter()
        rc = fn(img.ctypes.data, 3840, 2160, 3, 2.0, C.byref(o), C.byref(osz), None, None)
        ts.append(time.perf_counter() - t0); cs.append(time.process_time() - c0)
        assert rc == 0, L.srcnn_last_error()
        L.srcnn_delete_array(o)
    S.profile_enable(False)
    # device milliseconds per call and context (layer kernels + resampler, from the library's own events): a straggler --
    # a slow device, a share that fills its rounds badly -- shows as the odd one out
    per_ctx = []
    for k in range(n):
        pr = S.profile_read_context(k)
        per_ctx.append(round(sum(v[0] for v in pr.values()) / (args.reps + 1), 3))
    out["process_srcnn_4k_rgb_ms"] = {"best": round(min(ts[1:]) * 1e3, 2), "all": [round(t * 1e3, 2) for t in ts[1:]],
                                      "host_cpu_ms": round(sorted(cs[1:])[len(cs[1:]) // 2] * 1e3, 2),
                                      "device_ms_per_call_per_context": per_ctx}
    # (2) one 7680x4320 frame -> 15360x8640, tiled over the contexts vs the whole-frame call on context 0
    w, h = 7680, 4320
    y = synth.plane(h, w, synth.SEED0, "smooth")
    S.set_context(0)
    d_in = S.DeviceBuffer.from_numpy(y)
    d_out = S.DeviceBuffer(4 * w * h * 4)
    for name, call in (("whole_frame_call", lambda: (S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_out.ptr, None)), S.sync())),
                       ("node_tiled_4_sub_bands", lambda: S.check(L.srcnn_y_upscale2x_f32_node_dev(d_in.ptr, w, h, d_out.ptr, 4))),
                       ("node_tiled_1_sub_band", lambda: S.check(L.srcnn_y_upscale2x_f32_node_dev(d_in.ptr, w, h, d_out.ptr, 1)))):
        call()
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
        out[name + "_ms"] = round(min(ts) * 1e3, 2)
    # (3) host frame stream dealt over the contexts
    r = bench.pcie_inclusive(S, frames=8)
    out["host_stream_MPix_s"] = r["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
