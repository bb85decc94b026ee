"""CPU: the oracle (oracle/srcnn_oracle.c) against the committed golden vectors.

The golden vectors were produced by the REAL reference (tools/make_golden.py); the butterfly pair is
the reference's own published sample (Pictures/butterfly.png -> butterfly_srcnn.png and
butterfly_srcnn_convolution.png).  Integer/byte outputs and float planes alike must match bit for bit:
the oracle restates the same operation order and roundings (see its header).
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import ROOT, assert_bit_equal


def test_weights_blob(golden, oracle_lib):
    w = oracle_lib.weights()
    assert w.size == 8129
    assert np.array_equal(w.view(np.uint32), golden.weights.view(np.uint32))
    sha = hashlib.sha256(golden.weights.tobytes()).hexdigest()
    assert sha.startswith("822a078c") and sha.endswith("cfc699")


def test_butterfly_end_to_end(golden, oracle_lib):
    b = golden.butterfly
    rgb, conv = oracle_lib.process(b["rgb_in"], 2.0)
    assert np.array_equal(rgb, b["rgb_out"])
    assert np.array_equal(conv, b["conv_y"])
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == golden.known["butterfly"]["rgb_out_sha256"]
    assert hashlib.sha256(conv.tobytes()).hexdigest() == golden.known["butterfly"]["conv_y_sha256"]


PLANE_CASES = ["noise_24x40", "noise_29x37", "smooth_33x65", "row_1x17", "col_13x1", "tiny_2x3", "one_1x1",
               "noise_70x9", "smooth_7x130", "wild_12x16"]


@pytest.mark.parametrize("name", PLANE_CASES)
def test_y_planes(golden, oracle_lib, name):
    p = golden.planes
    out, up, c1, c2 = oracle_lib.y_path(p[name + "_in"], taps=True)
    assert_bit_equal(up, p[name + "_up"], name + " upscaled Y")
    assert_bit_equal(out, p[name + "_out"], name + " output")
    if name + "_c1" in p:
        assert_bit_equal(c1, p[name + "_c1"], name + " layer-1")
        assert_bit_equal(c2, p[name + "_c2"], name + " layer-2")


def test_constant_planes(golden, oracle_lib):
    for key, rec in golden.known["constant_planes"].items():
        out = oracle_lib.y_path(np.full((12, 16), float(key), np.float32))
        assert np.all(out.view(np.uint32) == rec["bits"]), (key, out[0, 0], rec)
    # size independence of the constant answer (every stage is position-invariant)
    big = oracle_lib.y_path(np.full((9, 31), 128.0, np.float32))
    assert np.all(big.view(np.uint32) == golden.known["constant_planes"]["128.0"]["bits"])


@pytest.mark.parametrize("filt", ["nearest", "bilinear", "bicubic", "lanczos3", "bspline"])
def test_resampler_filters_and_ratios(golden, oracle_lib, filt):
    r = golden.resample
    fid = ["nearest", "bilinear", "bicubic", "lanczos3", "bspline"].index(filt)
    for tag, (dw, dh) in (("x2", (46, 38)), ("x1p5", (34, 28)), ("x3", (69, 57)), ("down", (11, 9))):
        assert_bit_equal(oracle_lib.resample(r["in"], dw, dh, fid), r["%s_%s" % (filt, tag)], filt + tag)


def test_process_cases(golden, oracle_lib):
    p = golden.process
    out, conv = oracle_lib.process(p["rgba_in"], 2.0)
    assert np.array_equal(out, p["rgba_out"]) and np.array_equal(conv, p["rgba_conv"])
    for name, fid in (("nearest", 0), ("bilinear", 1), ("lanczos3", 3), ("bspline", 4)):
        out, conv = oracle_lib.process(p["rgb_in"], 2.0, fid)
        assert np.array_equal(out, p["rgb_%s_out" % name]), name
        assert np.array_equal(conv, p["rgb_%s_conv" % name]), name
    for tag, m in (("x15", 1.5), ("x3", 3.0)):
        out, conv = oracle_lib.process(p["rgb_in"], m)
        assert np.array_equal(out, p["rgb_%s_out" % tag]) and np.array_equal(conv, p["rgb_%s_conv" % tag])


def test_bicubic_2x_phase_structure(oracle_lib):
    """interior outputs of a 2x upscale use exactly two 4-tap phases."""
    left, right, w = oracle_lib.axis_table(74, 37)
    odd = [-0.0234375, 0.78211805555555558, 0.25607638888888884, -0.014756944444444444]
    for u in range(3, 2 * 37 - 3):
        taps = w[u, :right[u] - left[u] + 1]
        nz = taps[np.nonzero(taps)[0][0]:]
        assert len(nz) == 4
        ref = odd if u % 2 else odd[::-1]
        assert np.allclose(nz, ref, rtol=0, atol=1e-15), (u, nz)
    assert np.isclose(w[0, 0], 1.0308924485125861) and np.isclose(w[0, 1], -0.030892448512586119)


def test_translation_property_of_the_path(oracle_lib):
    """Size-independent property: away from the borders the path is translation-equivariant -- shifting the
    input by (dy, dx) shifts the 2x output by (2dy, 2dx), bit for bit (every stage is position-invariant;
    only the truncate-and-renormalise resampler borders and the clamp-to-edge padding break it)."""
    rng = np.random.default_rng(99)
    big = (rng.random((60, 72)) * 255).astype(np.float32)
    a = oracle_lib.y_path(big[:48, :56])
    b = oracle_lib.y_path(big[3:51, 5:61])             # same content shifted by (3, 5)
    m = 16                                            # > 6-px receptive field + resampler border zone
    assert np.array_equal(a[2 * 3 + m: 96 - m, 2 * 5 + m: 112 - m].view(np.uint32),
                          b[m: 96 - 2 * 3 - m, m: 112 - 2 * 5 - m].view(np.uint32))


def test_output_range_and_saturation(oracle_lib):
    """convolution55 clamps to [0,255] (src/libsrcnn.cpp:521-522): holds for wild inputs too."""
    rng = np.random.default_rng(3)
    y = ((rng.random((20, 28)) - 0.3) * 900).astype(np.float32)
    out = oracle_lib.y_path(y)
    assert out.min() >= 0.0 and out.max() <= 255.0
    assert (out == 0).any() and (out == 255).any()


def test_product_weight_table_matches_golden_blob(golden):
    """The PRODUCT's own weight image (libsrcnn_amd/csrc/srcnn_weights.inc, compiled into libsrcnn_amd.so) holds
    exactly the bit patterns of tests/golden/weights_f32.bin (= src/convdata.h in the order b1,W1,b2,W2,b3,W3),
    and so does the oracle's copy -- the two .inc files are byte-identical tables."""
    import re
    def parse(path):
        text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
        return np.array([int(t, 16) for t in re.findall(r"0x[0-9a-fA-F]+", text)], dtype=np.uint32)
    prod = parse(os.path.join(ROOT, "libsrcnn_amd", "csrc", "srcnn_weights.inc"))
    assert prod.size == 8129
    assert np.array_equal(prod, golden.weights.view(np.uint32))
    assert np.array_equal(parse(os.path.join(ROOT, "oracle", "oracle_weights.inc")), prod)
