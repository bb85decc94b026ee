"""Counter-based synthetic Y planes: any shard can generate its own frames.

`noise`  : uniform in [0,255) -- worst-case rounding, a few % of outputs saturate at 0/255.
`smooth` : low-frequency sinusoids + 8 % noise, image-like (<0.1 % saturation).
Seeds follow `0x5C0DE000 + frame_index`.
"""
import numpy as np

SEED0 = 0x5C0DE000


def _mix(x):
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xFF51AFD7ED558CCD)
        x = x ^ (x >> np.uint64(33)); x = x * np.uint64(0xC4CEB9FE1A85EC53)
        x = x ^ (x >> np.uint64(33))
    return x


def plane(h, w, seed, kind="noise"):
    idx = np.arange(h * w, dtype=np.uint64).reshape(h, w)
    with np.errstate(over="ignore"):
        x = _mix(idx + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15))
    u = (x >> np.uint64(40)).astype(np.float64) * 2.0 ** -24
    if kind == "noise":
        return (255.0 * u).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    s = 0.5 + 0.22 * np.sin(xx * 0.11 + 0.3 * (seed % 1000)) + 0.18 * np.cos(yy * 0.07 - xx * 0.013) + 0.08 * (u - 0.5)
    return (255.0 * np.clip(s, 0, 1)).astype(np.float32)


def frames(n, h, w, first_index=0, kind="smooth"):
    return np.stack([plane(h, w, SEED0 + first_index + i, kind) for i in range(n)])
