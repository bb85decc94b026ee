"""One RANK of a multi-process run of the library's RCCL path (tests/test_gpu_comm_ranks.py starts `world` of these).
All ranks share device 0 and talk through the RCCL stand-in (tests/rccl_double, loaded by the product via SRCNN_RCCL_LIB):
the installed RCCL refuses two ranks per device and the pool has one GPU per box.  The unique id travels through a file.
Prints one JSON object.  Test infrastructure."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import libsrcnn_amd as S          # noqa: E402
from libsrcnn_amd import synth, multigpu     # noqa: E402


def exchange_id(path, rank):
    ident = (C.c_ubyte * 128)()
    if rank == 0:
        S.check(S.lib().srcnn_comm_unique_id(ident))
        with open(path + ".tmp", "wb") as f:
            f.write(bytes(ident))
        os.rename(path + ".tmp", path)
        return ident
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 120:
            raise RuntimeError("rank 0 never published the id")
        time.sleep(0.01)
    return (C.c_ubyte * 128).from_buffer_copy(open(path, "rb").read())


def err():
    return S.lib().srcnn_last_error().decode("utf-8", "replace")


def cmd_tiled(rank, world, args):
    """srcnn_comm_tiled_y_upscale2x_f32_dev at world N: the frame gathered on the root == srcnn_y_upscale2x_f32_dev, bit for bit."""
    L = S.lib()
    res = {"rank": rank, "cases": []}
    for spec in args:                                  # "WxH:nsub:root"
        dims, nsub, root = spec.split(":")
        w, h = (int(v) for v in dims.split("x"))
        nsub, root = int(nsub), int(root)
        y = synth.plane(h, w, synth.SEED0 + w + h, "smooth" if w * h > 4_000_000 else "noise")
        d_in = S.DeviceBuffer.from_numpy(y)
        t = multigpu.TiledFrameGPU(w, h, rank, world, root=root, nsub=nsub)
        st = S.Stream()
        t0 = time.perf_counter()
        for _ in range(2):                             # twice: the second frame must wait for the first one's gathers
            t.step(d_in, st.handle)
        t.wait(st.handle)
        case = {"shape": [w, h], "nsub": nsub, "root": root, "ms_two_frames": round((time.perf_counter() - t0) * 1e3, 1)}
        if rank == root:
            got = t.result(st.handle)
            d_ref = S.DeviceBuffer(4 * w * h * 4)
            S.check(L.srcnn_y_upscale2x_f32_dev(d_in.ptr, w, h, d_ref.ptr, None))
            S.sync()
            ref = d_ref.to_numpy(np.float32, (2 * h, 2 * w))
            case["equal"] = bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))
            case["sha"] = hashlib.sha256(got.tobytes()).hexdigest()[:16]
            d_ref.free()
        S.check(L.srcnn_comm_barrier(None))            # nobody frees its band while the root may still be reading it
        res["cases"].append(case)
        st.destroy(); d_in.free(); t.d_band.free()
        if t.d_full:
            t.d_full.free()
    return res


def cmd_collectives(rank, world, args):
    """gather / gatherv (ragged, with an empty rank) / gatherv_at / allgather / barrier with known data."""
    L = S.lib()
    res = {"rank": rank}
    n = 1000
    mine = S.DeviceBuffer.from_numpy(np.full(n, rank + 1, np.float32))
    allb = S.DeviceBuffer(4 * n * world)
    S.check(L.srcnn_memset_dev(allb.ptr, 0, 4 * n * world, None)); S.sync()
    S.check(L.srcnn_comm_gather_f32(mine.ptr, n, allb.ptr, 0, None))
    S.check(L.srcnn_comm_wait(None))
    if rank == 0:
        got = allb.to_numpy(np.float32, (world, n))
        res["gather"] = bool(all((got[r] == r + 1).all() for r in range(world)))
    # ragged counts: rank r sends 100*r + 7 floats, the last rank nothing
    counts = [(100 * r + 7) if r < world - 1 or world == 1 else 0 for r in range(world)]
    c_arr = (C.c_size_t * world)(*counts)
    S.check(L.srcnn_memset_dev(allb.ptr, 0, 4 * n * world, None)); S.sync()
    S.check(L.srcnn_comm_gatherv_f32(mine.ptr, c_arr, allb.ptr, world - 1 if world > 1 else 0, None))     # root = last rank
    S.check(L.srcnn_comm_wait(None))
    if rank == (world - 1 if world > 1 else 0):
        got = allb.to_numpy(np.float32, (n * world,))
        pos, ok = 0, True
        for r in range(world):
            ok = ok and bool((got[pos:pos + counts[r]] == r + 1).all())
            pos += counts[r]
        res["gatherv"] = ok and not got[pos:].any()
    # allgather
    S.check(L.srcnn_comm_allgather_f32(mine.ptr, n, allb.ptr, None))
    S.check(L.srcnn_comm_wait(None))
    got = allb.to_numpy(np.float32, (world, n))
    res["allgather"] = bool(all((got[r] == r + 1).all() for r in range(world)))
    res["barrier"] = L.srcnn_comm_barrier(None)
    r_, n_ = C.c_int(-1), C.c_int(-1)
    S.check(L.srcnn_comm_rank(C.byref(r_), C.byref(n_)))
    res["rank_seen"], res["nranks_seen"] = r_.value, n_.value
    return res


def cmd_mismatch(rank, world, args):
    """The last rank asks for a different number of pieces: its gather tables differ.  The checksum all-reduce in front of the
    first gather (on by default) must make EVERY rank return SRCNN_E_COMM at once instead of pairing sends with the wrong
    receives."""
    L = S.lib()
    w, h = 640, 400
    d_in = S.DeviceBuffer.from_numpy(synth.plane(h, w, 3, "noise"))
    nsub = 3 if rank == world - 1 else 4
    t = multigpu.TiledFrameGPU(w, h, rank, world, nsub=nsub)
    t0 = time.perf_counter()
    rc = L.srcnn_comm_tiled_y_upscale2x_f32_dev(d_in.ptr, w, h, t.d_band.ptr, t.d_full.ptr if t.d_full else None, 0, nsub, None)
    return {"rank": rank, "rc": rc, "error": err(), "ms": round((time.perf_counter() - t0) * 1e3, 1),
            "check": os.environ.get("SRCNN_COMM_CHECK", "")}


def cmd_mismatch_later(rank, world, args):
    """every rank first agrees on table T1 (verified, cached); then the last rank alone comes with another table.
    The ranks that have T1 cached go straight to their send / receive while the last one all-reduces its new checksum: no
    pairing is possible, and what must happen is the DEADLINE -- every rank returns SRCNN_E_COMM after SRCNN_COMM_TIMEOUT_MS
    (set short by the test), the communicator is aborted, and destroy comes back."""
    L = S.lib()
    w, h = 640, 400
    d_in = S.DeviceBuffer.from_numpy(synth.plane(h, w, 3, "noise"))
    t = multigpu.TiledFrameGPU(w, h, rank, world, nsub=4)
    rc1 = L.srcnn_comm_tiled_y_upscale2x_f32_dev(d_in.ptr, w, h, t.d_band.ptr, t.d_full.ptr if t.d_full else None, 0, 4, None)
    rc1w = L.srcnn_comm_wait(None)
    nsub = 3 if rank == world - 1 else 4
    t0 = time.perf_counter()
    rc2 = L.srcnn_comm_tiled_y_upscale2x_f32_dev(d_in.ptr, w, h, t.d_band.ptr, t.d_full.ptr if t.d_full else None, 0, nsub, None)
    rc2w = L.srcnn_comm_wait(None) if rc2 == 0 else rc2
    ms = round((time.perf_counter() - t0) * 1e3, 1)
    e = err()
    t1 = time.perf_counter()
    rc_destroy = L.srcnn_comm_destroy()
    return {"rank": rank, "rc1": rc1, "rc1w": rc1w, "rc2": rc2, "rc2w": rc2w, "error": e, "ms": ms, "rc_destroy": rc_destroy,
            "destroy_ms": round((time.perf_counter() - t1) * 1e3, 1)}


def cmd_missing(rank, world, args):
    """The last rank never makes the call (it sleeps, then leaves).  Every other rank must come back with SRCNN_E_COMM once
    the deadline (SRCNN_COMM_TIMEOUT_MS, set short by the test) has passed -- and be able to destroy the communicator."""
    L = S.lib()
    w, h = 640, 400
    d_in = S.DeviceBuffer.from_numpy(synth.plane(h, w, 3, "noise"))
    t = multigpu.TiledFrameGPU(w, h, rank, world, nsub=2)
    if rank == world - 1:
        time.sleep(float(args[0]) if args else 4.0)
        return {"rank": rank, "absent": True}
    t0 = time.perf_counter()
    rc = L.srcnn_comm_tiled_y_upscale2x_f32_dev(d_in.ptr, w, h, t.d_band.ptr, t.d_full.ptr if t.d_full else None, 0, 2, None)
    rc_wait = L.srcnn_comm_wait(None) if rc == 0 else rc
    ms = round((time.perf_counter() - t0) * 1e3, 1)
    e = err()
    rc_after = L.srcnn_comm_barrier(None)
    t1 = time.perf_counter()
    rc_destroy = L.srcnn_comm_destroy()
    return {"rank": rank, "rc": rc, "rc_wait": rc_wait, "error": e, "ms": ms, "rc_after": rc_after,
            "rc_destroy": rc_destroy, "destroy_ms": round((time.perf_counter() - t1) * 1e3, 1)}


def cmd_destroy_with_dead_peer(rank, world, args):
    """the tiled call returns once everything is QUEUED.  A rank whose peer then dies must still get out of
    srcnn_comm_destroy (it used to sit in a bare hipDeviceSynchronize for ever): the drain is bounded by the deadline and the
    communicator is aborted on a miss."""
    L = S.lib()
    w, h = 640, 400
    d_in = S.DeviceBuffer.from_numpy(synth.plane(h, w, 3, "noise"))
    t = multigpu.TiledFrameGPU(w, h, rank, world, nsub=1)
    S.check(L.srcnn_comm_barrier(None))
    if rank == 0:                                      # the root "dies": it never posts its receives
        time.sleep(float(args[0]) if args else 4.0)
        return {"rank": rank, "absent": True}
    counts = (C.c_size_t * world)(*([16] * world))
    offs = (C.c_size_t * world)(*[16 * r for r in range(world)])
    # table verification needs every rank: switch it off by going through the raw gather with SRCNN_COMM_CHECK=0 (set by the test)
    rc = L.srcnn_comm_gatherv_at_f32(d_in.ptr, counts, offs, None, 0, None)         # a send to the root, queued and returned
    t0 = time.perf_counter()
    rc_destroy = L.srcnn_comm_destroy()
    return {"rank": rank, "rc": rc, "rc_destroy": rc_destroy, "destroy_ms": round((time.perf_counter() - t0) * 1e3, 1)}


def main():
    cmd, rank, world, idfile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    args = sys.argv[5:]
    # one device per rank when the test asks for the REAL RCCL on a multi-GPU box; device 0 for everybody behind the stand-in
    S.init(rank % max(1, S.device_count()) if os.environ.get("SRCNN_WORKER_DEVICE_PER_RANK") else 0)
    ident = exchange_id(idfile, rank)
    S.check(S.lib().srcnn_comm_init(ident, rank, world))
    out = globals()["cmd_" + cmd](rank, world, args)
    if not out.get("absent") and "rc_destroy" not in out:
        S.lib().srcnn_comm_destroy()
    print(json.dumps(out), flush=True)
    os._exit(0) if out.get("absent") else None         # the absent rank leaves without tearing anything down, like a crash


if __name__ == "__main__":
    main()
