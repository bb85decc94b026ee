import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, libsrcnn_amd as S
S.init(0); L = S.lib()
img = bench.synth_rgb(2160, 3840, 0x5C0DE000 + 2160)
S.ConfigureFilterSRCNN(S.SRCNNF_Bicubic, False)
out = np.empty((4320, 7680, 3), np.uint8)
def call():
    S.check(L.srcnn_process_u8(img.ctypes.data, 3840, 2160, 3, 2.0, 2, out.ctypes.data, None))
for pause in (0.0, 0.002, 0.01, 0.05):
    call(); call()
    S.profile_reset(); S.profile_enable(True)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
        if pause: time.sleep(pause)
    S.profile_enable(False)
    p = S.profile_read()
    print("pause %.3f s: wall best %.2f ms median %.2f | per call: %s" % (pause, min(ts) * 1e3, sorted(ts)[4] * 1e3, {k: round(v[0] / 8, 3) for k, v in p.items()}))
