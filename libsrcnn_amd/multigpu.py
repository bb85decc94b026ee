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


def gather_offsets(counts):
    """Where rank r's counts[r] elements land in the root's contiguous buffer (what srcnn_comm_gatherv_f32 does):
    the exclusive prefix sums of the per-rank counts."""
    offs, pos = [], 0
    for c in counts:
        offs.append(pos)
        pos += c
    return offs


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


def plan(world, frames_per_rank=4, in_w=3840, in_h=2160, tiled_w=7680, tiled_h=4320, devices=None):
    """Dry run of both shardings for `world` ranks: what every rank would own and allocate, checked for
    consistency WITHOUT touching a device (tools/run_8gpu.sh --dry-run, tests/test_multi_gpu_cpu.py).
    Returns a dict; raises AssertionError if the partition is not exact."""
    devices = world if devices is None else devices
    out_h, out_w = 2 * tiled_h, 2 * tiled_w
    ranks = []
    pos = 0
    for r in range(world):
        row0, rows = band_rows(out_h, r, world)
        assert row0 == pos, (r, row0, pos)
        pos += rows
        lo, hi = band_input_rows(row0, rows, tiled_h)
        ca, cb = max(0, row0 - 2), min(out_h, row0 + rows + 2)
        ua, ub = max(0, ca - 4), min(out_h, cb + 4)
        ranks.append({
            "rank": r, "device": r % max(1, devices),
            "frames": {"first_index": r * frames_per_rank, "count": frames_per_rank,
                       "seeds": ["0x%08X" % (0x5C0DE000 + r * frames_per_rank + f) for f in range(frames_per_rank)],
                       "in_bytes": frames_per_rank * in_w * in_h * 4, "out_bytes": frames_per_rank * 4 * in_w * in_h * 4,
                       "scratch_bytes": 128 * 4 * in_w * in_h + 4 * 4 * in_w * in_h + 4 * in_w * 2 * in_h},
            "band": {"row0": row0, "rows": rows, "input_rows": [lo, hi], "count_floats": rows * out_w,
                     "offset_floats": row0 * out_w, "band_bytes": rows * out_w * 4,
                     "scratch_bytes": 128 * out_w * (cb - ca) + 4 * out_w * (ub - ua) + 4 * tiled_w * (ub - ua)},
        })
    assert pos == out_h
    counts = [rk["band"]["count_floats"] for rk in ranks]
    assert sum(counts) == out_h * out_w
    assert [rk["band"]["offset_floats"] for rk in ranks] == [sum(counts[:r]) for r in range(world)]
    all_seeds = [s for rk in ranks for s in rk["frames"]["seeds"]]
    assert len(set(all_seeds)) == len(all_seeds)
    return {"world": world, "devices": devices, "ranks": ranks, "gather_counts": counts,
            "tiled_frame": {"in": [tiled_w, tiled_h], "out": [out_w, out_h], "root_bytes": out_h * out_w * 4}}


# ---------------------------------------------------------------------------------------------------
# GPU-side pieces (only imported/used where a device exists)
# ---------------------------------------------------------------------------------------------------
def init_comm_from_torch_dist(dist, rank, world):
    """Create the RCCL communicator inside libsrcnn_amd.so, shipping the unique id over an already
    initialised torch.distributed (gloo) group (world == 1: no group needed)."""
    import ctypes as C
    import libsrcnn_amd as S
    ident = (C.c_ubyte * 128)()
    if rank == 0:
        S.check(S.lib().srcnn_comm_unique_id(ident))
    if world > 1:
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0)
        ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
    S.check(S.lib().srcnn_comm_init(ident, rank, world))


def gpu_compute_band(d_in, w, h, row0, rows, d_band, stream=None):
    import libsrcnn_amd as S
    S.check(S.lib().srcnn_y_upscale2x_f32_band_dev(d_in.ptr, w, h, row0, rows, d_band.ptr, stream))


class TiledFrameGPU:
    """The device side of upscale2x_frame_tiled, one process per GPU: srcnn_comm_tiled_y_upscale2x_f32_dev computes this
    rank's band in `nsub` pieces and gathers piece k (RCCL peer->root ncclSend/ncclRecv, explicit offsets, ragged bands
    allowed) on the library's comm stream while piece k+1 computes, into a contiguous (2h x 2w) frame on the root.
    Buffers are allocated once; step() is what bench.py times (the caller synchronises)."""

    def __init__(self, w, h, rank, world, root=0, nsub=4):
        import ctypes as C
        import libsrcnn_amd as S
        self.S, self.w, self.h, self.rank, self.world, self.root = S, w, h, rank, world, root
        self.nsub = max(1, int(nsub))
        r0, rows = C.c_uint(0), C.c_uint(0)
        S.check(S.lib().srcnn_band_rows(2 * h, rank, world, C.byref(r0), C.byref(rows)))
        self.row0, self.rows = r0.value, rows.value
        assert (self.row0, self.rows) == band_rows(2 * h, rank, world)      # the C partition is the one the CPU tests cover
        self.d_band = S.DeviceBuffer(max(1, self.rows) * 2 * w * 4)
        self.d_full = S.DeviceBuffer(4 * w * h * 4) if rank == root else None

    def step(self, d_in, stream=None):
        S, L = self.S, self.S.lib()
        S.check(L.srcnn_comm_tiled_y_upscale2x_f32_dev(d_in.ptr, self.w, self.h, self.d_band.ptr,
                                                       self.d_full.ptr if self.d_full else None, self.root, self.nsub, stream))

    def wait(self, stream=None):
        """Bounded host wait for the frame queued by step() on `stream` (srcnn_comm_wait: SRCNN_E_COMM instead of a hang when a
        peer is missing), on every rank."""
        self.S.check(self.S.lib().srcnn_comm_wait(stream))

    def result(self, stream=None):
        """Host copy of the assembled frame (root only).  Waits through srcnn_comm_wait first: a plain device sync would hang
        for ever behind a receive whose peer died."""
        self.wait(stream)
        self.S.sync()
        return self.d_full.to_numpy(np.float32, (2 * self.h, 2 * self.w))
