"""CPU: the oracle against the real reference compiled from its own sources (oracle/_ref).

oracle/_ref/libsrcnn_ref.so exists only if `make -C oracle ref` ran where /root/reference is present
(the dev container); it travels to the GPU box as a built artefact.  Skipped when absent -- the golden
vectors in tests/golden (made from the same library) still pin the oracle.
"""
import numpy as np
import pytest

import oracle
from conftest import assert_bit_equal

pytestmark = pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def ref():
    return oracle.Reference()


def test_weights_identical(ref, oracle_lib):
    assert np.array_equal(ref.weights().view(np.uint32), oracle_lib.weights().view(np.uint32))


@pytest.mark.parametrize("shape,seed", [((24, 40), 0), ((31, 17), 1), ((1, 9), 2), ((9, 1), 3), ((2, 2), 4),
                                        ((48, 6), 5)])
def test_stagewise_bit_identical(ref, oracle_lib, shape, seed):
    rng = np.random.default_rng(seed)
    y = (rng.random(shape) * 255).astype(np.float32)
    got = oracle_lib.y_path(y, taps=True)
    want = ref.y_path(y, taps=True)
    for g, w, what in zip(got, want, ("out", "upscaled", "layer-1", "layer-2")):
        assert_bit_equal(g, w, what)


def test_layers_fed_independently(ref, oracle_lib):
    rng = np.random.default_rng(7)
    y = (rng.random((14, 22)) * 255).astype(np.float32)
    assert_bit_equal(oracle_lib.conv1(y), ref.conv1(y), "conv1")
    c1 = (rng.random((64, 10, 12)) * 300 - 20).astype(np.float32)
    assert_bit_equal(oracle_lib.conv2(c1), ref.conv2(c1), "conv2")
    c2 = (rng.random((32, 11, 13)) * 200).astype(np.float32)
    assert_bit_equal(oracle_lib.conv3(c2), ref.conv3(c2), "conv3")


@pytest.mark.parametrize("filt", range(5))
@pytest.mark.parametrize("dst", [(46, 38), (30, 20), (23, 40), (40, 19), (8, 7)])
def test_resampler(ref, oracle_lib, filt, dst):
    rng = np.random.default_rng(filt)
    y = (rng.random((19, 23)) * 255).astype(np.float32)
    assert_bit_equal(oracle_lib.resample(y, dst[0], dst[1], filt), ref.resample(y, dst[0], dst[1], filt))


def test_process_rgb_rgba(ref, oracle_lib):
    rng = np.random.default_rng(11)
    for d in (3, 4):
        img = rng.integers(0, 256, (17, 21, d), dtype=np.uint8)
        for m, f in ((2.0, 2), (1.5, 2), (2.0, 1), (2.0, 0), (2.5, 3)):
            a = oracle_lib.process(img, m, f)
            b = ref.process(img, m, f)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (d, m, f)
