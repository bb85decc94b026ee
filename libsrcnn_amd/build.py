"""Build libsrcnn_amd.so (gfx950 code object + host C ABI + C++ drop-in) with hipcc, in-tree.

Usage: python -m libsrcnn_amd.build [--force]
The output lands in libsrcnn_amd/lib/ so that it travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsrcnn_amd.so")

SOURCES = ["srcnn_kernels.hip", "srcnn_fused_f16.hip", "srcnn_capi.cpp", "srcnn_pipeline.cpp", "srcnn_comm.cpp", "dropin.cpp"]
DEPS = SOURCES + ["../../tools/srcnntest.cpp", "srcnn_kernels.h", "srcnn_host.hpp", "resample_table.hpp", "srcnn_weights.inc",
                  "../../include/srcnn_amd.h", "../../include/libsrcnn_dropin.h"]

# -ffp-contract=off: strict kernels and the host table builder must round every multiply and add
# separately (the reference binary contains no FMA).  FAST kernels call fmaf explicitly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-result", "-Wno-unused-value", "-Wno-ignored-attributes", "-D__HIP_PLATFORM_AMD__"]


# per-source extras.  The fused kernel is issue-bound beside its MFMAs, where packed fp32 VALU ops are slower than the
# scalar pair they replace (MI355X_MICROARCH.md, cycle constants): keep the SLP vectoriser from forming them.
EXTRA_FLAGS = {"srcnn_fused_f16.hip": ["-fno-slp-vectorize"] + os.environ.get("SRCNN_FUSED_CFLAGS", "").split()}


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


STAMP = os.path.join(LIBDIR, "build.sha256")


def source_digest():
    """sha256 over the CONTENT of every input of the build (sources, headers, this recipe, the flags): a prebuilt .so that
    travelled with a snapshot can therefore never mask sources that changed under it, whatever the file times say."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr((FLAGS, sorted(EXTRA_FLAGS.items()))).encode())
    for d in sorted(set(DEPS)) + [os.path.abspath(__file__)]:
        path = d if os.path.isabs(d) else os.path.join(CSRC, d)
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def stale():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    try:
        return open(STAMP).read().strip() != source_digest()
    except OSError:
        return True


def kernel_source_sha(name="k_conv12_mfma"):
    """sha256 of the source text of one kernel (from its template header to the next banner comment): profiles that quote a
    counter for that kernel record it, and bench.py refuses to quote a counter taken from a different kernel text."""
    import hashlib
    import re
    src = open(os.path.join(CSRC, "srcnn_kernels.hip")).read()
    m = re.search(r"template <[^>]*>\s*__global__[^\n]*void %s\(" % re.escape(name), src)
    if not m:
        return None
    end = src.find("// ====", m.end())
    return hashlib.sha256(src[m.start():end if end > 0 else len(src)].encode()).hexdigest()


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc()] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    build_cli(verbose)
    with open(STAMP, "w") as f:
        f.write(source_digest() + "\n")
    return LIB


def build_cli(verbose=True):
    """tools/srcnntest.cpp -> libsrcnn_amd/bin/srcnntest: the counterpart of the reference's CLI harness,
    linked against the drop-in library exactly as a libsrcnn user would link."""
    root = os.path.dirname(HERE)
    bindir = os.path.join(HERE, "bin")
    os.makedirs(bindir, exist_ok=True)
    exe = os.path.join(bindir, "srcnntest")
    cmd = ["g++", "-O2", "-std=c++17", os.path.join(root, "tools", "srcnntest.cpp"), "-L" + LIBDIR, "-lsrcnn_amd",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,$ORIGIN/../lib", "-o", exe]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return exe


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
