#!/usr/bin/env python3
"""A/B builds: tools/build_variant.py NAME [-DX=Y ...] -> build/ab/NAME.so

Recompiles srcnn_kernels.hip with the extra defines (tuning macros of an experiment, e.g. -DSRCNN_C3_MC=2) and links it
with the product's other objects (libsrcnn_amd/lib/*.o, built by `python -m libsrcnn_amd.build`).  The result is loaded
with SRCNN_AMD_LIB=... (tools/lib_ab.py a.so b.so).  build/ is git-ignored and travels to the GPU box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libsrcnn_amd import build as B          # noqa: E402

name, defs = sys.argv[1], sys.argv[2:]
out_dir = os.path.join(ROOT, "build", "ab")
os.makedirs(out_dir, exist_ok=True)
B.build(verbose=False)
obj = os.path.join(out_dir, name + ".kernels.o")
subprocess.check_call([B.hipcc()] + B.FLAGS + defs + ["-x", "hip", "-c", os.path.join(B.CSRC, "srcnn_kernels.hip"), "-o", obj])
objs = [obj] + [os.path.join(B.LIBDIR, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s != "srcnn_kernels.hip"]
lib = os.path.join(out_dir, name + ".so")
subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib, "-ldl", "-Wl,-soname,libsrcnn_amd.so",
                       "-Wl,--version-script=" + os.path.join(B.CSRC, "exports.map")])
print(lib)
