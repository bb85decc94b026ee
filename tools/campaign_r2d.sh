#!/bin/bash
O=gpurun_out/r2d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 ./tools/ubench/strict_tap > $O/strict_tap.txt 2>&1
cat $O/strict_tap.txt
python - > $O/import_order.txt 2>&1 <<'PY'
import subprocess, sys
for order in ("lib_first", "torch_first"):
    code = ("import sys; sys.path.insert(0,'.');\n"
            + ("import libsrcnn_amd as S; S.init(0); import torch, torch.distributed\n" if order == "lib_first" else "import torch, torch.distributed; import libsrcnn_amd as S; S.init(0)\n")
            + "import numpy as np, ctypes as C\nfrom libsrcnn_amd import synth, multigpu\n"
              "multigpu.init_comm_from_torch_dist(None, 0, 1)\n"
              "y = synth.plane(40, 60, 1, 'noise'); t = multigpu.TiledFrameGPU(60, 40, 0, 1); d = S.DeviceBuffer.from_numpy(y); t.step(d); S.sync()\n"
              "ok = bool(np.array_equal(t.result(), S.y_upscale2x(y)))\n"
              "S.check(S.lib().srcnn_comm_destroy())\n"
              "maps = [l.split()[-1] for l in open('/proc/self/maps') if ('libamdhip64' in l or 'librccl' in l)]\n"
              "print(sorted(set(maps)), 'gather_ok', ok)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    print(order, "rc", r.returncode, r.stdout.strip()[-700:], "|", r.stderr.strip()[-300:])
PY
cat $O/import_order.txt
