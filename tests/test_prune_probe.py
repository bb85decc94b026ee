"""CPU: why the non-parity tier does NOT prune layer-1 channels (SURVEY.md 8f-4 suggested it).

15 of the 64 layer-1 filters have max|w| < 0.004 (src/convdata.h:32-674), so their activations are almost the constant
relu(b1[k]).  Replacing them by that constant was measured here with the oracle's own layer-1 activations and an fp64
evaluation of layers 2+3: the output moves by 2.6e-3 (butterfly) .. 5e-3 (noise) on the 0..255 scale -- 5-10x the
tier's 5e-4 bound and 20x the reference's own rounding noise -- and it would not even save matrix instructions
(64 -> 49 channels still needs two 32-channel MFMA blocks).  This test pins that measurement so the decision is
reproducible; a 16th channel is not remotely prunable (|dY| of several units)."""
import numpy as np

from libsrcnn_amd import synth


def _layers23_f64(w, c1):
    b2 = w[5248:5280]; W2 = w[5280:5280 + 2048].reshape(32, 64); b3 = w[7328]; W3 = w[7329:].reshape(32, 5, 5)
    c2 = np.maximum(np.tensordot(W2, c1, axes=(1, 0)) + b2[:, None, None], 0)
    H, W = c2.shape[1:]
    p = np.pad(c2, ((0, 0), (2, 2), (2, 2)), mode="edge")
    out = np.full((H, W), b3)
    for dy in range(5):
        for dx in range(5):
            out += np.tensordot(W3[:, dx, dy], p[:, dy:dy + H, dx:dx + W], axes=(0, 0))     # W3[m][x][y], x = column
    return np.clip(out, 0, 255)


def test_dead_channel_pruning_exceeds_the_fast_tier_bound(oracle_lib):
    w = oracle_lib.weights().astype(np.float64)
    b1 = w[:64]; W1 = w[64:64 + 5184].reshape(64, 81)
    order = np.argsort(np.abs(W1).max(1))
    assert np.abs(W1).max(1)[order[14]] < 0.004 < 0.1 < np.abs(W1).max(1)[order[15]]      # exactly 15 near-dead filters
    y = synth.plane(72, 72, synth.SEED0 + 1, "noise")
    out, _up, c1, _c2 = oracle_lib.y_path(y, taps=True)
    c1 = c1.astype(np.float64)
    base = _layers23_f64(w, c1)
    assert np.abs(base - out).max() < 3e-4                       # the fp64 evaluation is the reference up to its rounding noise
    pruned = c1.copy()
    for k in order[:15]:
        pruned[k] = max(b1[k], 0.0)
    err = np.abs(_layers23_f64(w, pruned) - base).max()
    assert 1e-3 < err < 2e-2, err                                # ~5e-3: an order of magnitude over the 5e-4 bound
