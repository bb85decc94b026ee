"""TEST INFRASTRUCTURE: a stand-in for librccl that lets several processes sharing ONE GPU run the library's multi-rank code
(see rccl_double.cpp).  build() compiles it into tests/rccl_double/_build/ (git-ignored; travels to the GPU box with the
snapshot like the product's own .so); the product loads it through its SRCNN_RCCL_LIB hook only when a test says so."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rccl_double.cpp")
LIB = os.path.join(HERE, "_build", "librccl_double.so")
SYMBOLS = ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv",
           "ncclAllGather", "ncclAllReduce", "ncclGetErrorString", "ncclCommAbort"]


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    rocm = os.environ.get("ROCM_PATH") or "/opt/rocm"
    # host code only (HIP runtime API, no kernels): plain g++ against the same libamdhip64 the product is linked to
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-Wall", "-I" + rocm + "/include",
                           SRC, "-o", LIB, "-L" + rocm + "/lib", "-Wl,-rpath," + rocm + "/lib", "-lamdhip64", "-lpthread", "-lrt"])
    return LIB
