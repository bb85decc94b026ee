#!/usr/bin/env python3
"""Probe: does the D2H of frame i overlap the kernels of frame i+1?  Two compute streams (slots) as in
srcnn_y_upscale2x_f32_stream, with the D2H either on the slot's own stream or on a separate copy stream behind an event."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, libsrcnn_amd as S
from libsrcnn_amd import synth
S.init(0); L = S.lib()
w, h, F = 3840, 2160, 8
nin, nout = w*h*4, 4*w*h*4
pin_in = L.srcnn_host_alloc_pinned(F*nin); pin_out = L.srcnn_host_alloc_pinned(F*nout)
fr = np.ctypeslib.as_array(C.cast(pin_in, C.POINTER(C.c_float)), (F, h, w))
fr[:] = synth.plane(h, w, 0, "smooth")[None]
st = [S.Stream(), S.Stream()]; cp = [S.Stream(), S.Stream()]
din = [S.DeviceBuffer(nin) for _ in range(2)]; dout = [S.DeviceBuffer(nout) for _ in range(2)]
ev_k = [S.Event(), S.Event()]; ev_c = [S.Event(), S.Event()]
def run(mode):
    t0 = time.perf_counter()
    for f in range(F):
        s = f & 1
        S.check(L.srcnn_memcpy_h2d(din[s].ptr, pin_in + f*nin, nin, st[s].handle))
        S.check(L.srcnn_y_upscale2x_f32_dev(din[s].ptr, w, h, dout[s].ptr, st[s].handle))
        if mode == "same":
            S.check(L.srcnn_memcpy_d2h(pin_out + f*nout, dout[s].ptr, nout, st[s].handle))
        elif mode == "hostsync":          # host waits for the kernels, then the copy goes to a copy-only stream
            st[s].sync()
            S.check(L.srcnn_memcpy_d2h(pin_out + f*nout, dout[s].ptr, nout, cp[s].handle))
        else:                             # device-side dependency: event on the kernel stream, copy stream waits for it
            ev_k[s].record(st[s])
            S.check(L.srcnn_stream_wait_event(cp[s].handle, ev_k[s].handle))
            S.check(L.srcnn_memcpy_d2h(pin_out + f*nout, dout[s].ptr, nout, cp[s].handle))
            ev_c[s].record(cp[s])
            S.check(L.srcnn_stream_wait_event(st[s].handle, ev_c[s].handle))
    for x in st + cp: x.sync()
    return time.perf_counter() - t0
for mode in ("same", "hostsync", "gpuwait"):
    run(mode)
    t = min(run(mode) for _ in range(3))
    print("%-10s %.1f ms per %d frames = %.2f ms/frame = %.0f MPix/s" % (mode, t*1e3, F, t*1e3/F, F*4*w*h/1e6/t))
# compute only / copies only
def only(kind):
    t0 = time.perf_counter()
    for f in range(F):
        s = f & 1
        if kind == "k": S.check(L.srcnn_y_upscale2x_f32_dev(din[s].ptr, w, h, dout[s].ptr, st[s].handle))
        elif kind == "d2h": S.check(L.srcnn_memcpy_d2h(pin_out + f*nout, dout[s].ptr, nout, st[s].handle))
        else: S.check(L.srcnn_memcpy_h2d(din[s].ptr, pin_in + f*nin, nin, st[s].handle))
    for x in st: x.sync()
    return (time.perf_counter() - t0) / F * 1e3
for kind in ("k", "d2h", "h2d"):
    only(kind); print("only %-4s %.2f ms/frame" % (kind, min(only(kind) for _ in range(3))))
