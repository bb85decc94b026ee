#!/bin/bash
O=gpurun_out/r2c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q --maxfail=10 -p no:cacheprovider -k "fast or fused or canaries or unfused or f16" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
# which HIP / RCCL does a process end up with, depending on import order (torch bundles its own ROCm 7.0 libs)?
python - > $O/import_order.txt 2>&1 <<'PY'
import subprocess, sys
for order in ("lib_first", "torch_first"):
    code = ("import sys; sys.path.insert(0,'.');\n"
            + ("import libsrcnn_amd as S; S.init(0); import torch, torch.distributed\n" if order == "lib_first" else "import torch, torch.distributed; import libsrcnn_amd as S; S.init(0)\n")
            + "import numpy as np, ctypes as C\nfrom libsrcnn_amd import synth, multigpu\n"
              "multigpu.init_comm_from_torch_dist(None, 0, 1)\n"
              "y = synth.plane(40, 60, 1, 'noise'); a = S.y_upscale2x(y)\n"
              "S.check(S.lib().srcnn_comm_destroy())\n"
              "maps = [l.split()[-1] for l in open('/proc/self/maps') if ('libamdhip64' in l or 'librccl' in l or 'libhsa-runtime' in l)]\n"
              "print(sorted(set(maps)), float(a.sum()))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    print(order, r.returncode, r.stdout.strip()[-600:], r.stderr.strip()[-300:])
PY
cat $O/import_order.txt
bash tools/collect_profiles.sh r02a > $O/collect.log 2>&1
tail -3 $O/collect.log
