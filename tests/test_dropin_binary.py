"""Binary-level drop-in (the reference's artefacts: lib/libsrcnn.a by default, libsrcnn.so + an installed
libsrcnn.h, /root/reference/Makefiles/Makefile.linux:13-14,38-39,64-75).  An application is built ONCE, with -lsrcnn, against
the reference's own shared library (oracle/_ref, the reference compiled from its sources, under the name its Makefile gives
it) and the SAME binary is then run on this library by pointing LD_LIBRARY_PATH at libsrcnn_amd/lib: no relinking.
CPU part: return codes.  GPU part: the butterfly image, bytes equal to the reference's published pair."""
import hashlib
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "libsrcnn_amd", "lib")
REF = os.path.join(ROOT, "oracle", "_ref", "libsrcnn_ref.so")
APP = os.path.join(ROOT, "tests", "dropin_app.cpp")


def _product():
    from libsrcnn_amd import build
    if build.stale():
        build.build(verbose=False)


@pytest.fixture(scope="module")
def app(tmp_path_factory):
    """(binary, directory holding the REFERENCE under the name libsrcnn.so).  Header: the one `make install` installs as
    libsrcnn.h (the reference's own where the reference tree is present -- both must work)."""
    import oracle
    _product()
    if not os.path.exists(REF):
        oracle.build(ref=True)
    if not os.path.exists(REF):
        pytest.skip("no compiled reference here (oracle/_ref)")
    td = tmp_path_factory.mktemp("dropin")
    refdir = td / "ref"; refdir.mkdir()
    shutil.copy(REF, refdir / "libsrcnn.so")
    inc = td / "inc"; inc.mkdir()
    if os.path.isdir("/root/reference/src"):
        shutil.copy("/root/reference/src/libsrcnn.h", inc / "libsrcnn.h")
    else:
        shutil.copy(os.path.join(ROOT, "include", "libsrcnn_dropin.h"), inc / "libsrcnn.h")
    exe = td / "app"
    subprocess.check_call(["g++", "-std=c++11", "-O1", APP, "-I" + str(inc), "-L" + str(refdir), "-lsrcnn", "-fopenmp", "-o", str(exe)])
    needed = subprocess.run(["readelf", "-d", str(exe)], capture_output=True, text=True).stdout
    assert "libsrcnn.so" in needed and "srcnn_amd" not in needed
    return str(exe), str(refdir)


def run(exe, libdir, *args):
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""), OMP_NUM_THREADS="4")
    r = subprocess.run([exe] + [str(a) for a in args], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.split()


def test_same_binary_runs_on_the_reference_and_on_this_library(app):
    exe, refdir = app
    ref = run(exe, refdir)
    assert ref == ["-1", "-2", "0", str(8 * 8 * 3)], ref          # the reference itself
    ours = run(exe, LIBDIR)                                       # the same binary, no relink
    assert ours[:2] == ["-1", "-2"], ours
    assert ours[2] in ("0", "-200"), ours                         # -200: no gfx950 device here (there is no CPU path)
    if ours[2] == "0":
        assert ours[3] == str(8 * 8 * 3)


def test_libsrcnn_so_is_the_product_under_the_reference_name():
    _product()
    assert os.path.samefile(os.path.join(LIBDIR, "libsrcnn.so"), os.path.join(LIBDIR, "libsrcnn_amd.so"))
    d = subprocess.run(["readelf", "-d", os.path.join(LIBDIR, "libsrcnn.so")], capture_output=True, text=True).stdout
    assert "soname: [libsrcnn_amd.so]" in d, d


def test_static_archive_and_install_layout(tmp_path):
    """libsrcnn.a (the reference's default artefact) links with the documented line, and `make install` lays out what the
    reference's install target does: the libraries and a libsrcnn.h a program can compile against."""
    _product()
    dest = tmp_path / "root"
    subprocess.check_call(["make", "-s", "install", "DESTDIR=" + str(dest), "PREFIX=/usr"], cwd=ROOT)
    lib, inc = dest / "usr" / "lib", dest / "usr" / "include"
    assert sorted(os.listdir(lib)) == ["libsrcnn.a", "libsrcnn.so", "libsrcnn_amd.so"]
    assert sorted(os.listdir(inc)) == ["libsrcnn.h", "srcnn_amd.h"]
    exe = tmp_path / "static_app"
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["g++", "-std=c++11", APP, "-I" + str(inc), str(lib / "libsrcnn.a"), "-L" + rocm + "/lib",
                           "-Wl,-rpath," + rocm + "/lib", "-lamdhip64", "-ldl", "-lpthread", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = out.stdout.split()
    assert got[:2] == ["-1", "-2"] and got[2] in ("0", "-200"), got
    dyn = tmp_path / "dyn_app"                                     # and against the installed shared pair, by the reference's name
    subprocess.check_call(["g++", "-std=c++11", APP, "-I" + str(inc), "-L" + str(lib), "-lsrcnn", "-Wl,-rpath," + str(lib), "-o", str(dyn)])
    got = subprocess.run([str(dyn)], capture_output=True, text=True, timeout=300).stdout.split()
    assert got[:2] == ["-1", "-2"] and got[2] in ("0", "-200"), got
    subprocess.check_call(["make", "-s", "uninstall", "DESTDIR=" + str(dest), "PREFIX=/usr"], cwd=ROOT)
    assert os.listdir(lib) == [] and os.listdir(inc) == []


@pytest.mark.gpu
def test_same_binary_butterfly_bytes_equal_the_golden_pair(app, golden, tmp_path):
    """GPU: the binary linked against the reference, run on this library, reproduces Pictures/butterfly_srcnn.png and
    butterfly_srcnn_convolution.png byte for byte (sha256 from tests/golden/known_answers.json)."""
    exe, _ = app
    g = golden.butterfly
    src = tmp_path / "in.rgb"
    np.ascontiguousarray(g["rgb_in"]).tofile(src)
    out, conv = tmp_path / "out.rgb", tmp_path / "conv.y"
    got = run(exe, LIBDIR, src, 256, 256, 3, 2.0, out, conv)
    assert got == ["0", str(512 * 512 * 3), str(512 * 512)], got
    known = golden.known["butterfly"]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == known["rgb_out_sha256"]
    assert hashlib.sha256(open(conv, "rb").read()).hexdigest() == known["conv_y_sha256"]
