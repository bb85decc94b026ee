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
DEPS = SOURCES + ["../../tools/srcnntest.cpp", "srcnn_kernels.h", "srcnn_host.hpp", "srcnn_settings.hpp", "srcnn_watchdog.hpp", "resample_table.hpp", "srcnn_weights.inc",
                  "../../include/srcnn_amd.h", "../../include/srcnn_amd_debug.h", "../../include/libsrcnn_dropin.h", "exports.map"]

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
    if not all(os.path.exists(p) for p in (LIB, STAMP, os.path.join(LIBDIR, "libsrcnn.so"), os.path.join(LIBDIR, "libsrcnn.a"))):
        return True
    try:
        return open(STAMP).read().strip() != source_digest()
    except OSError:
        return True


def _region(src, start_pat, end_pat):
    import re
    m = re.search(start_pat, src)
    if not m:
        return None
    e = re.search(end_pat, src[m.end():])
    return src[m.start():m.end() + (e.start() if e else len(src) - m.end())]


def kernel_source_sha(name="k_conv12_mfma"):
    """sha256 of everything that decides what one launch of a kernel moves through HBM: profiles that quote a counter for
    that kernel record it, and bench.py refuses to quote a counter taken from a different text.  Covered: the kernel's body
    (from its template header to the next banner comment) and, for the layer kernels, the constants its tile and LDS geometry
    are built from (M_* / m_* for k_conv12_mfma, C3_* for k_conv3), the LDS-DMA primitive rs_dma_dword they stage through, the
    launchers that set grid, block and dynamic LDS (launch_conv12_mfma / conv12_grid_info, launch_conv3) and the
    host function that picks the variant (run_conv12 in srcnn_capi.cpp)."""
    import hashlib
    import re
    src = open(os.path.join(CSRC, "srcnn_kernels.hip")).read()
    m = re.search(r"template <[^>]*>\s*__global__[^\n]*void %s\(" % re.escape(name), src)
    if not m:
        return None
    end = src.find("// ====", m.end())
    parts = [src[m.start():end if end > 0 else len(src)]]
    extra = []
    if name == "k_conv12_mfma":
        extra = [_region(src, r"constexpr int M_NW\b", r"\n// One tap-step"),
                 _region(src, r"__device__ __forceinline__ void rs_dma_dword\(", r"\n}\n"),
                 _region(src, r"void conv12_grid_info\(", r"\n}\n"),
                 _region(src, r"void launch_conv12_mfma\(", r"\n}\n"),
                 _region(open(os.path.join(CSRC, "srcnn_capi.cpp")).read(), r"void run_conv12\(", r"\n}\n")]
    elif name == "k_conv3":
        extra = [_region(src, r"constexpr int C3_TW\b", r"\ntemplate <bool STRICT"),
                 _region(src, r"__device__ __forceinline__ void rs_dma_dword\(", r"\n}\n"),
                 _region(src, r"void launch_conv3\(", r"\n}\n")]
    if any(e is None for e in extra):
        return None                      # a region moved: better no fingerprint than a partial one
    h = hashlib.sha256()
    for part in parts + extra:
        h.update(part.encode())
        h.update(b"\0")
    return h.hexdigest()


STRICT_DIR = os.path.join(LIBDIR, "strict")
STRICT_LIB = os.path.join(STRICT_DIR, "libsrcnn_amd.so")


def build_strict_only(force=False, verbose=True):
    """The strict-only product (`make STRICT_ONLY=1`): -DSRCNN_STRICT_ONLY, no srcnn_fused_f16.hip -- not one instance of a
    non-parity kernel (FAST, FAST_F16, RELAXED) is compiled, srcnn_set_mode refuses every mode but SRCNN_MODE_STRICT with
    SRCNN_E_UNSUPPORTED, and the exported symbol set is the full library's.  Lands in libsrcnn_amd/lib/strict/ (load it with
    SRCNN_AMD_LIB=...).  Returns (path, seconds spent compiling or None if it was up to date)."""
    import time
    stamp = os.path.join(STRICT_DIR, "build.sha256")
    digest = source_digest() + "-strict"
    if not force and os.path.exists(STRICT_LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        return STRICT_LIB, None
    os.makedirs(STRICT_DIR, exist_ok=True)
    t0 = time.time()
    objs = []
    for src in SOURCES:
        if src == "srcnn_fused_f16.hip":
            continue
        obj = os.path.join(STRICT_DIR, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc()] + FLAGS + ["-DSRCNN_STRICT_ONLY"] + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", STRICT_LIB, "-ldl", "-Wl,-soname,libsrcnn_amd.so",
                                                                          "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(digest + "\n")
    return STRICT_LIB, time.time() - t0


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
    # exports.map: only srcnn_* and the two reference symbols are exported (not the kernels' launch stubs)
    cmds = [[hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB, "-ldl", "-Wl,-soname,libsrcnn_amd.so",
                                                                             "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")],
            # the names the reference's Makefiles produce (Makefiles/Makefile.linux:13-14,38-39): libsrcnn.so is a symbolic
            # link to the product (old binaries ask the loader for that file name), libsrcnn.a the same objects as an archive
            ["ln", "-sf", "libsrcnn_amd.so", os.path.join(LIBDIR, "libsrcnn.so")],
            ["rm", "-f", os.path.join(LIBDIR, "libsrcnn.a")],
            ["ar", "crs", os.path.join(LIBDIR, "libsrcnn.a")] + objs]
    for cmd in cmds:
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
    if "--strict-only" in sys.argv:
        lib, secs = build_strict_only(force="--force" in sys.argv)
        print(lib, "(%.0f s, %d bytes)" % (secs, os.path.getsize(lib)) if secs is not None else "(up to date, %d bytes)" % os.path.getsize(lib))
    else:
        build(force="--force" in sys.argv)
        print(LIB)
