"""CPU: the C-ABI library loads, exports every symbol include/*.h declares, refuses to compute without
a device, and its host-side table builder agrees with the oracle.  No GPU compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def S():
    import libsrcnn_amd as S
    if not os.path.exists(S.LIB_PATH):
        from libsrcnn_amd import build
        build.build(verbose=False)
    return S


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srcnn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(S):
    """The stable ABI is frozen: include/srcnn_amd.h declares exactly the functions of the committed list include/srcnn_amd.abi
    (and so does the binding), the instruments live in include/srcnn_amd_debug.h, and the library exports both sets.  Adding a
    function to the stable header without editing the list (and bumping SRCNN_AMD_ABI_VERSION) fails here."""
    names = _declared_functions("srcnn_amd.h")
    assert len(names) >= 40
    frozen = [ln.strip() for ln in open(os.path.join(ROOT, "include", "srcnn_amd.abi")) if ln.strip() and not ln.startswith("#")]
    assert frozen == sorted(frozen) and len(set(frozen)) == len(frozen)
    assert names == frozen, "stable header and include/srcnn_amd.abi disagree: %s" % (set(names) ^ set(frozen))
    assert sorted(S.STABLE_ABI_SYMBOLS) == frozen, set(S.STABLE_ABI_SYMBOLS) ^ set(frozen)
    debug = [n for n in _declared_functions("srcnn_amd_debug.h") if n not in names]
    assert sorted(S.DEBUG_SYMBOLS) == debug, set(S.DEBUG_SYMBOLS) ^ set(debug)
    assert not any(n.startswith("srcnn_debug_") or n in ("srcnn_fused_diag", "srcnn_set_relaxation") for n in names), "an instrument in the stable header"
    header = open(os.path.join(ROOT, "include", "srcnn_amd.h")).read()
    assert "#define SRCNN_AMD_ABI_VERSION 5" in header and "SRCNN_MODE_RELAXED" not in header
    L = S.lib()
    missing = [n for n in names + debug if not hasattr(L, n)]
    assert not missing, missing


def test_reference_cxx_symbols_exported(S):
    """The reference's own mangled names (nm -D of its libsrcnn.so)."""
    L = S.lib()
    for sym in S.CXX_SYMBOLS:
        assert hasattr(L, sym), sym
    assert L.srcnn_abi_version() == 5


def test_exports_are_exactly_the_abi(S):
    """nm -D --defined-only: the C ABI of include/srcnn_amd.h plus the two symbols of the reference's src/libsrcnn.h:46-54, and
    nothing else (csrc/exports.map; the HIP kernels' launch stubs used to leak out although the TUs are -fvisibility=hidden)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", S.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert names == sorted(S.C_ABI_SYMBOLS + S.CXX_SYMBOLS), set(names) ^ set(S.C_ABI_SYMBOLS + S.CXX_SYMBOLS)


def test_settings_are_one_table_and_design_md_carries_it(S):
    """Every SRCNN_* switch lives in csrc/srcnn_settings.hpp, is read once, and is printable (srcnn_debug_settings, no device);
    DESIGN.md section 6 is generated from the same table (tools/gen_settings_table.py) and must carry exactly its rows."""
    import subprocess
    import sys
    text = S.debug_settings()
    names = [ln.split("=")[0] for ln in text.splitlines()]
    assert len(names) >= 20 and len(set(names)) == len(names) and all(n.startswith("SRCNN_") for n in names)
    # no launcher keeps a private getenv: the only one in the product is the table reader
    src = "".join(open(os.path.join(ROOT, "libsrcnn_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "libsrcnn_amd", "csrc"))
                  if f.endswith((".cpp", ".hip", ".hpp", ".h")))
    assert src.count("getenv(") == 1
    # a child with switches set reports them; the parent's table is unaffected (read at load time)
    code = "import sys; sys.path.insert(0, %r); import libsrcnn_amd as S; print(S.debug_settings())" % ROOT
    child = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SRCNN_CONV12_DMA="0", SRCNN_MAX_LANES="99", SRCNN_BANDS="0.5"),
                           capture_output=True, text=True, check=True).stdout
    assert "SRCNN_CONV12_DMA=0 (default 1)" in child and "SRCNN_MAX_LANES=64 (default 4)" in child and "SRCNN_BANDS=0.5" in child
    rows = S.debug_settings(markdown=True).strip().splitlines()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    missing = [r for r in rows if r not in design]
    assert not missing, "DESIGN.md section 6 is stale: run tools/gen_settings_table.py\n" + "\n".join(missing[:3])


def test_reference_argument_checks_need_no_device(S):
    """ProcessSRCNN's early returns (src/libsrcnn.cpp:951-966) happen before any device work."""
    img = np.zeros((4, 4, 3), np.uint8)
    assert S.ProcessSRCNN(None, 4, 4, 3, 2.0)[0] == -1
    assert S.ProcessSRCNN(img, 0, 4, 3, 2.0)[0] == -1
    assert S.ProcessSRCNN(img, 4, 0, 3, 2.0)[0] == -1
    assert S.ProcessSRCNN(img, 4, 4, 0, 2.0)[0] == -1
    assert S.ProcessSRCNN(img, 4, 4, 3, 0.0)[0] == -2
    assert S.ProcessSRCNN(img, 4, 4, 3, -1.0)[0] == -2


def test_round4_settings_need_no_device(S):
    """The mode / relaxation mask and the comm deadline are plain settings (no device touched): values are validated, setters
    return the previous value, the asynchronous ProcessSRCNN pair and the bounded comm wait fail cleanly without a device /
    communicator, and the registry behind the page-locked fast path knows only blocks the library handed out."""
    import ctypes as C
    L = S.lib()
    assert L.srcnn_get_mode() == S.MODE_STRICT
    assert L.srcnn_set_relaxation(S.RELAX_L1 | S.RELAX_L3_X64) == S.RELAX_L3_X64           # default mask: exact layer-3 products
    assert L.srcnn_set_mode(S.MODE_RELAXED) == S.MODE_STRICT and L.srcnn_get_mode() == S.MODE_RELAXED
    assert L.srcnn_set_relaxation(0) == (S.RELAX_L1 | S.RELAX_L3_X64) and L.srcnn_get_mode() == S.MODE_RELAXED
    assert L.srcnn_set_relaxation(16) < 0 and L.srcnn_set_relaxation(S.RELAX_L3_X64 | S.RELAX_L3_F32) < 0
    assert L.srcnn_set_mode(7) < 0
    assert L.srcnn_set_mode(S.MODE_STRICT) == S.MODE_RELAXED
    assert L.srcnn_set_relaxation(S.RELAX_L3_X64) == 0
    prev = L.srcnn_comm_set_timeout_ms(1234)
    assert prev == int(__import__("os").environ.get("SRCNN_COMM_TIMEOUT_MS", 60000))
    assert L.srcnn_comm_set_timeout_ms(prev) == 1234 and L.srcnn_comm_set_timeout_ms(-5) < 0
    assert L.srcnn_comm_wait(None) == -204                                                 # no communicator
    if S.device_count() == 0:
        job = C.c_void_p()
        buf = np.zeros(64, np.uint8)
        assert L.srcnn_process_u8_begin(buf.ctypes.data, 2, 2, 3, 2.0, 2, buf.ctypes.data, None, C.byref(job)) == -200
        assert not job.value
        assert not L.srcnn_host_alloc_pinned(4096)
        assert L.srcnn_debug_clock_probe(1) == -200


def test_no_cpu_fallback(S):
    if S.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(S.SrcnnError) as e:
        S.y_upscale2x(np.zeros((4, 4), np.float32))
    assert e.value.code == -200
    rc, out, conv = S.ProcessSRCNN(np.zeros((4, 4, 3), np.uint8), 4, 4, 3, 2.0)
    assert rc == -200 and out is None


@pytest.mark.parametrize("dst,src,filt", [(16, 8, 2), (74, 37, 2), (18, 9, 2), (10, 5, 2), (2, 1, 2), (4, 2, 2),
                                          (6, 3, 2), (46, 23, 1), (34, 23, 3), (69, 23, 4), (9, 19, 0), (11, 19, 2),
                                          (2160, 1080, 2), (57, 19, 1)])
def test_axis_table_matches_oracle(S, oracle_lib, dst, src, filt):
    a = S.axis_table(dst, src, filt)
    b = oracle_lib.axis_table(dst, src, filt)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for u in range(dst):
        n = a[1][u] - a[0][u] + 1
        assert np.array_equal(a[2][u, :n].view(np.uint64), b[2][u, :n].view(np.uint64)), u


def test_axis_table_property_random_sizes(S, oracle_lib):
    """Property test: for random (filter, destination length, source length) -- up-scales, down-scales, 1-pixel axes, prime
    ratios -- the PRODUCT's host-side table builder (csrc/resample_table.hpp, through srcnn_axis_table: no device involved)
    produces the oracle's bounds and weights bit for bit, and every range stays inside the source."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=150, deadline=None)
    @given(filt=st.integers(0, 4), dst=st.integers(1, 700), src=st.integers(1, 700))
    def check(filt, dst, src):
        a = S.axis_table(dst, src, filt)
        b = oracle_lib.axis_table(dst, src, filt)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert a[0].min() >= 0 and a[1].max() <= src - 1 and np.all(a[1] >= a[0])
        n = a[1] - a[0] + 1
        cols = np.arange(a[2].shape[1])[None, :]
        used = cols < n[:, None]
        assert np.array_equal(a[2].view(np.uint64)[used], b[2].view(np.uint64)[used])

    check()


def test_output_size_matches_reference_geometry(S, golden):
    """Geometry of the golden ProcessSRCNN outputs (incl. step scaling, made by the real reference)."""
    p = golden.process
    assert S.output_size(256, 256, 2.0) == (512, 512)
    h, w = p["rgb_in"].shape[:2]
    assert S.output_size(w, h, 1.5) == p["rgb_x15_out"].shape[1::-1]
    assert S.output_size(w, h, 3.0) == p["rgb_x3_out"].shape[1::-1]
    assert S.output_size(24, 20, 4.0, True) == p["rgb_x4step_out"].shape[1::-1]
    assert S.output_size(24, 20, 3.0, True) == p["rgb_x3step_out"].shape[1::-1]
    with pytest.raises(S.SrcnnError):
        S.output_size(0, 4, 2.0)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference header not present on this machine")
def test_program_built_against_reference_header_links_and_runs(S, tmp_path):
    """A C++ program compiled against the REFERENCE's own libsrcnn.h (default argument, reference parameters,
    enum) links against libsrcnn_amd.so unchanged and gets the reference's early return codes."""
    import subprocess
    src = tmp_path / "app.cpp"
    src.write_text(r'''
#include <cstdio>
#include "libsrcnn.h"
int main() {
    ConfigureFilterSRCNN(SRCNNF_Lanczos3);            // default stepscale argument from the reference header
    ConfigureFilterSRCNN(SRCNNF_Bicubic, false);
    unsigned char px[4 * 4 * 3] = {0};
    unsigned char* out = nullptr; unsigned outsz = 0;
    int a = ProcessSRCNN(nullptr, 4, 4, 3, 2.0f, out, outsz, nullptr, nullptr);
    int b = ProcessSRCNN(px, 4, 4, 3, -1.0f, out, outsz, nullptr, nullptr);
    int c = ProcessSRCNN(px, 4, 4, 3, 2.0f, out, outsz, nullptr, nullptr);   // 0 on a GPU box, -200 without a device
    std::printf("%d %d %d %u\n", a, b, c, outsz);
    delete[] out;
    return 0;
}
''')
    exe = tmp_path / "app"
    libdir = os.path.dirname(S.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", str(src), "-I/root/reference/src", "-L" + libdir, "-lsrcnn_amd",
                           "-Wl,-rpath," + libdir, "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120).stdout.split()
    assert out[0] == "-1" and out[1] == "-2"
    assert out[2] in ("0", "-200")
    if out[2] == "0":
        assert out[3] == str(8 * 8 * 3)


def test_process_band_plan_partitions_and_honours_the_workspace_limit(S):
    """srcnn_process_u8's band cut points (host logic, no device): they partition the row range in order, interior cuts are
    multiples of 16 rows, and no band holds more layer-2 scratch (128 B per output pixel) than srcnn_set_workspace_limit
    allows -- except the documented 16-row floor."""
    L = S.lib()
    cuts = (C.c_uint * 4096)()
    prev = L.srcnn_set_workspace_limit(16 << 30)
    try:
        for limit in (16 << 30, 1 << 30, 64 << 20, 1 << 20):
            L.srcnn_set_workspace_limit(limit)
            for (r0, r1, dw, many) in [(0, 4320, 7680, 0), (0, 540, 7680, 1), (1072, 2160, 3840, 1), (0, 300, 800, 0), (0, 8640, 15360, 0), (5, 6, 9, 0)]:
                n = L.srcnn_debug_band_plan(r0, r1, dw, many, cuts, 4096)
                assert 2 <= n <= 4096
                c = list(cuts[:n])
                assert c[0] == r0 and c[-1] == r1 and all(a < b for a, b in zip(c, c[1:]))
                cap_rows = max(16, limit // (128 * dw) - 4)
                for a, b in zip(c, c[1:]):
                    assert b - a <= cap_rows, (limit, r0, r1, dw, a, b, cap_rows)
                if r1 - r0 >= 512:
                    assert n >= 4                      # large ranges are always pipelined over several bands
    finally:
        L.srcnn_set_workspace_limit(prev)
    assert L.srcnn_debug_band_plan(10, 10, 100, 0, cuts, 16) < 0


def test_process_band_plan_random_ranges(S):
    """The same invariants over 600 random (row range, width, budget, lone / one-of-many) combinations, incl. the seven-band
    plan of large lone shares and its SRCNN_BANDS-free defaults: ordered partition, every band within the budget (16-row
    floor), a first band no larger than a third of a large range (the GPU must start early) and a short last band."""
    import numpy as np
    L = S.lib()
    rng = np.random.default_rng(20261003)
    cuts = (C.c_uint * 8192)()
    prev = L.srcnn_set_workspace_limit(16 << 30)
    try:
        for _ in range(600):
            dw = int(rng.integers(1, 16000))
            r0 = int(rng.integers(0, 5000))
            r1 = r0 + int(rng.integers(1, 9000))
            many = int(rng.integers(0, 2))
            limit = int(rng.choice([16 << 30, 2 << 30, 256 << 20, 8 << 20]))
            L.srcnn_set_workspace_limit(limit)
            n = L.srcnn_debug_band_plan(r0, r1, dw, many, cuts, 8192)
            assert 2 <= n <= 8192, (r0, r1, dw, many, limit, n)
            c = list(cuts[:n])
            assert c[0] == r0 and c[-1] == r1 and all(a < b for a, b in zip(c, c[1:])), c
            cap_rows = max(16, limit // (128 * dw) - 4)
            assert all(b - a <= cap_rows for a, b in zip(c, c[1:])), (limit, dw, c)
            if r1 - r0 >= 2048 and dw * (r1 - r0) >= 3000000 and limit == 16 << 30 and dw * (r1 - r0) * 128 < limit:
                assert c[1] - c[0] <= (r1 - r0) // 3 + 16, c          # early start
                assert c[-1] - c[-2] <= (r1 - r0) // 4 + 16, c        # short tail
    finally:
        L.srcnn_set_workspace_limit(prev)


def test_strict_only_build_has_the_same_abi_and_no_non_parity_kernel(S):
    """`make STRICT_ONLY=1` / `python -m libsrcnn_amd.build --strict-only`: the same exported symbols as the full library, not one
    non-parity kernel in the code object (no FAST / FAST_F16 / RELAXED instance), and every mode but STRICT refused with
    SRCNN_E_UNSUPPORTED -- a setting, so this needs no device."""
    import subprocess
    import sys
    from libsrcnn_amd import build
    strict, _ = build.build_strict_only(verbose=False)

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported(strict) == exported(S.LIB_PATH)
    assert os.path.getsize(strict) < os.path.getsize(S.LIB_PATH)
    blob = open(strict, "rb").read()
    full = open(S.LIB_PATH, "rb").read()
    for kernel in (b"k_fused_f16", b"k_conv3_fast", b"k_conv12_mfmaILi3", b"k_conv12_mfmaILi1", b"k_conv12_mfmaILi2"):
        assert kernel in full and kernel not in blob, kernel
    assert b"k_conv12_mfmaILi0ELb1" in blob and b"k_conv3ILb1ELb0ELb0ELb1" in blob           # the production strict kernels
    code = ("import sys; sys.path.insert(0, %r); import libsrcnn_amd as S; L = S.lib();"
            "print(L.srcnn_abi_version(), L.srcnn_set_mode(1), L.srcnn_set_mode(2), L.srcnn_set_mode(3), L.srcnn_set_mode(0), L.srcnn_get_mode(), L.srcnn_set_mode(9))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SRCNN_AMD_LIB=strict), capture_output=True, text=True, check=True)
    assert r.stdout.split() == ["5", "-203", "-203", "-203", "0", "0", "-1"], r.stdout


def test_rccl_lib_override_is_refused_under_secure_execution(S, tmp_path):
    """SRCNN_RCCL_LIB makes the library dlopen a file named by the environment: a trust boundary.  Like
    LD_PRELOAD it must not work in a process the kernel marked secure (AT_SECURE: set-uid / set-gid / file capabilities).
    Needs root and a mount that honours set-uid bits to stage such a process; skipped otherwise."""
    import shutil
    import subprocess
    if os.geteuid() != 0 or not shutil.which("gcc"):
        pytest.skip("needs root (to hand a set-uid binary to another user) and gcc")
    src = tmp_path / "secure_probe.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include <sys/auxv.h>\n#include "srcnn_amd_debug.h"\n'
                   'int main(void) { static char buf[16384]; srcnn_debug_settings(buf, sizeof buf, 0);\n'
                   '  char* p = strstr(buf, "SRCNN_RCCL_LIB="); char* e = p ? strchr(p, \' \') : 0; if (e) *e = 0;\n'
                   '  printf("%lu|%s\\n", getauxval(AT_SECURE), p ? p : "?"); return 0; }\n')
    import tempfile
    # the probe runs as `nobody`: library and binary go where that user can reach them (not under /root, not under pytest's 0700 tmp)
    stage = tempfile.mkdtemp(prefix="srcnn_secure_", dir="/tmp")
    try:
        os.chmod(stage, 0o755)
        shutil.copy(S.LIB_PATH, os.path.join(stage, "libsrcnn_amd.so"))
        exe = os.path.join(stage, "secure_probe")
        subprocess.run(["gcc", str(src), "-I", os.path.join(ROOT, "include"), "-L", stage, "-lsrcnn_amd", "-Wl,-rpath," + stage, "-o", exe], check=True)
        os.chown(exe, 65534, 65534)
        os.chmod(exe, 0o4755)
        env = dict(os.environ, SRCNN_RCCL_LIB="/nonexistent/librccl_from_the_environment.so")
        plain = subprocess.run([exe], env=dict(env, SRCNN_RCCL_LIB=""), capture_output=True, text=True)
        r = subprocess.run([exe], env=env, capture_output=True, text=True)
    finally:
        shutil.rmtree(stage, ignore_errors=True)
    if r.returncode != 0 or not r.stdout.startswith("1|"):
        pytest.skip("this mount does not honour set-uid bits (AT_SECURE stayed 0) or the probe could not run: " + r.stdout + r.stderr)
    assert r.stdout.strip() == "1|SRCNN_RCCL_LIB=(unset)", r.stdout                  # not honoured ...
    assert "SRCNN_RCCL_LIB=/nonexistent/librccl_from_the_environment.so ignored: secure-execution" in r.stderr      # ... and said so
    assert plain.stdout.strip() == "1|SRCNN_RCCL_LIB=(unset)" and "ignored" not in plain.stderr
    # the same probe in an ordinary process honours the variable
    code = "import sys; sys.path.insert(0, %r); import libsrcnn_amd as S; print([l for l in S.debug_settings().splitlines() if 'RCCL_LIB' in l][0])" % ROOT
    import sys
    o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    assert o.startswith("SRCNN_RCCL_LIB=/nonexistent/librccl_from_the_environment.so "), o


def test_switches_that_are_not_understood_are_said_once(S):
    """A retired or misspelt SRCNN_* switch and a value that does not parse are reported on stderr when the library is loaded
    (an A/B script that still exports a round-4 name would otherwise measure the production path and call it a variant);
    the words people write for booleans are understood."""
    import subprocess
    import sys
    code = "import sys; sys.path.insert(0, %r); import libsrcnn_amd as S; print(S.debug_settings())" % ROOT
    env = dict(os.environ, SRCNN_CONV12_VARIANT="2", SRCNN_TRACE="yes", SRCNN_MAX_LANES="many", SRCNN_THP="off", SRCNN_TEST_SEED="5",
               SRCNN_BENCH_FAKE="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True)
    assert "SRCNN_CONV12_VARIANT is not a switch of this library" in r.stderr
    assert "SRCNN_MAX_LANES=many is not an integer" in r.stderr
    assert "SRCNN_TEST_SEED" not in r.stderr and "SRCNN_BENCH_FAKE" not in r.stderr and "SRCNN_TRACE" not in r.stderr
    assert "SRCNN_TRACE=1 (default 0)" in r.stdout and "SRCNN_THP=0 (default 1)" in r.stdout and "SRCNN_MAX_LANES=4 (default 4)" in r.stdout
