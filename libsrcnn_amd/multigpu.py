"""One-process-per-GPU sharding of the SRCNN Y path (host logic; no device code here).

Two ways the path shards:
  * frames   -- independent; rank r owns frames r, r+N, ...  No collective at all.
  * one frame-- the OUTPUT is cut into horizontal bands, one per rank.  Every band is computed from the
                source frame (all halo comes from the input), so there is no exchange during compute;
                the only collective is the final gather of bands (RCCL over xGMI on the GPU path,
                srcnn_comm_gather_f32: peer->root sends, one link per band).

The functions take the band computer and the gather as callables so the same logic runs on the GPU
(libsrcnn_amd C ABI + RCCL) and in the CPU tests (oracle + gloo).
"""
import numpy as np


def shard_frames(n_frames, rank, world):
    """Frame indices owned by `rank` (round-robin, so a stream can be sharded without knowing its length)."""
    return list(range(rank, n_frames, world))


def band_rows(out_h, rank, world):
    """(row0, rows) of rank's band of the out_h output rows: contiguous, ordered by rank, sizes differ by
    at most one row, earlier ranks take the remainder."""
    base, rem = divmod(out_h, world)
    rows = base + (1 if rank < rem else 0)
    row0 = rank * base + min(rank, rem)
    return row0, rows


def band_input_rows(row0, rows, in_h, margin=8):
    """Input rows a 2x band depends on: +-2 rows of layer-2 activations, +-4 rows of upscaled Y for those,
    4-tap resampler (+-2 input rows) -> within +-6 output rows = +-5 input rows.  `margin` >= 5."""
    lo = max(0, row0 // 2 - margin)
    hi = min(in_h, (row0 + rows + 1) // 2 + margin)
    return lo, hi


def upscale2x_frame_tiled(y, rank, world, compute_band, gather):
    """Rank-local part of the tiled single-frame path.
    compute_band(y, row0, rows) -> (rows, 2w) float32 band;
    gather(band, counts) -> on rank 0 the list of all ranks' bands (None elsewhere)."""
    h, w = y.shape
    row0, rows = band_rows(2 * h, rank, world)
    band = compute_band(y, row0, rows) if rows > 0 else np.empty((0, 2 * w), np.float32)
    counts = [band_rows(2 * h, r, world)[1] for r in range(world)]
    parts = gather(band, counts)
    if parts is None:
        return None
    return np.concatenate([p.reshape(-1, 2 * w) for p in parts], axis=0)


# ---------------------------------------------------------------------------------------------------
# GPU-side pieces (only imported/used where a device exists)
# ---------------------------------------------------------------------------------------------------
def init_comm_from_torch_dist(dist, rank, world):
    """Create the RCCL communicator inside libsrcnn_amd.so, shipping the unique id over an already
    initialised torch.distributed (gloo) group."""
    import ctypes as C
    import libsrcnn_amd as S
    ident = (C.c_ubyte * 128)()
    if rank == 0:
        S.check(S.lib().srcnn_comm_unique_id(ident))
    box = [bytes(ident)]
    dist.broadcast_object_list(box, src=0)
    ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
    S.check(S.lib().srcnn_comm_init(ident, rank, world))


def gpu_compute_band(d_in, w, h, row0, rows, d_band, stream=None):
    import libsrcnn_amd as S
    S.check(S.lib().srcnn_y_upscale2x_f32_band_dev(d_in.ptr, w, h, row0, rows, d_band.ptr, stream))
