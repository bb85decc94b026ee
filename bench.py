#!/usr/bin/env python3
"""bench.py -- SRCNN Y-channel 2x throughput on MI355X (the metric BASELINE.json names).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload ("step"): one pass of the hot path -- 2x Mitchell upscale + conv 9x9x1->64 + conv 1x1x64->32 +
conv 5x5x32->1, strict (bit-exact) mode -- over a batch of F synthetic 3840x2160 planar-float32 Y frames
per GPU, producing F 7680x4320 frames.  Inputs and outputs are resident in HBM for the whole timed
region.  Frames are independent, so ranks share nothing on the data path: each rank owns its own frames
(weak scaling); torch.distributed (gloo) is used only for the barriers and the max-over-ranks of the time.

One JSON line on rank 0.  `value` = output megapixels of ALL ranks / max-over-ranks wall time.
`roofline` is for the dominant kernel (layers 1+2, k_conv12_mfma): its average launch duration is measured
live inside the timed region with HIP events recorded on the launch stream by the library
(srcnn_profile_*), and priced with the algorithmic FLOPs per launch.
`cpu_baseline` = the reference's own OpenMP path (oracle/_ref, compiled from the reference sources) or the
C restatement (oracle/) timed on this host's cores on a bounded sample of the same kind of frame.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IN_W, IN_H = 3840, 2160                   # "4K" input frame; output 7680x4320
MAC_L12 = 64 * 81 + 32 * 64               # 7232 MAC per output pixel in the dominant kernel
MAC_ALL = MAC_L12 + 32 * 25               # 8032
PEAK_F32_TFLOPS = 157.3                   # MI355X_MICROARCH.md: FP32 matrix == FP32 vector peak
PEAK_HBM_GBS = 8000.0


def cpu_baseline(S, budget_s=20.0):
    """The CPU-baseline leg (the only place bench.py touches oracle/): time the reference CPU path on this
    host -- a calibration frame first, then the largest frame that fits the budget -- and, on that same
    frame, compare the GPU output with the CPU reference's (the "max |dY| vs CPU ref" half of the metric).
    Returns (cpu_baseline object, max_abs_dY)."""
    import oracle
    from libsrcnn_amd import synth
    threads = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    if oracle.have_reference():
        eng, kind = oracle.Reference(), "reference"
    else:
        eng, kind = oracle.Oracle(), "port"
    cases = [(270, 480), (540, 960), (1080, 1920), (2160, 3840)]
    best = None
    spent = 0.0
    for h, w in cases:
        y = synth.plane(h, w, synth.SEED0, "smooth")
        t0 = time.perf_counter()
        ref_out = eng.y_path(y)
        dt = time.perf_counter() - t0
        spent += dt
        best = (h, w, dt, y, ref_out)
        if spent + dt * 4.2 > budget_s:      # the next size is 4x the pixels
            break
    h, w, dt, y, ref_out = best
    gpu_out = S.y_upscale2x(y)
    max_abs = float(np.max(np.abs(gpu_out.astype(np.float64) - ref_out.astype(np.float64))))
    obj = {"value": round(4 * h * w / 1e6 / dt, 4), "unit": "MPix/s", "cores": threads, "kind": kind,
           "sample": "1 synthetic %dx%d -> %dx%d Y frame, %.2f s wall, OMP_NUM_THREADS=%s (layer 1 can use at most 64 "
                     "threads, layer 2 at most 32: src/libsrcnn.cpp:791,817); GPU output of the same frame compared "
                     "element-wise" % (w, h, 2 * w, 2 * h, dt, os.environ["OMP_NUM_THREADS"])}
    return obj, max_abs


def side_workload(args):
    """The other BASELINE.json configurations.  Same timing protocol; reported with the same keys but they are
    NOT the headline line the driver records (that is --workload frames)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import libsrcnn_amd as S
    from libsrcnn_amd import synth, multigpu
    S.init(local_rank % max(1, S.device_count()))
    L = S.lib()

    def barrier():
        if dist is not None:
            dist.barrier()

    if args.workload == "tiled8k":
        w, h = 7680, 4320
        y = synth.plane(h, w, synth.SEED0, "smooth")
        d_in = S.DeviceBuffer.from_numpy(y)
        row0, rows = multigpu.band_rows(2 * h, rank, world)
        maxrows = multigpu.band_rows(2 * h, 0, world)[1]
        d_band = S.DeviceBuffer(maxrows * 2 * w * 4)
        d_full = S.DeviceBuffer(world * maxrows * 2 * w * 4) if rank == 0 else S.DeviceBuffer(16)
        multigpu.init_comm_from_torch_dist(dist, rank, world) if world > 1 else None
        if world == 1:
            import ctypes as C
            ident = (C.c_ubyte * 128)()
            S.check(L.srcnn_comm_unique_id(ident)); S.check(L.srcnn_comm_init(ident, 0, 1))

        def step():
            S.check(L.srcnn_y_upscale2x_f32_band_dev(d_in.ptr, w, h, row0, rows, d_band.ptr, None))
            S.check(L.srcnn_comm_gather_f32(d_band.ptr, maxrows * 2 * w, d_full.ptr, 0, None))
        mpix_step = 4 * w * h / 1e6
        label = "one 7680x4320 Y frame -> 15360x8640, %d output bands + RCCL gather to rank 0" % world
    elif args.workload == "host-stream":
        w, h, F = 3840, 2160, max(args.frames, 4)
        import ctypes as C
        F = max(F, 16)
        # caller-side page-locked frame buffers (what a capture/playout pipeline would hand over)
        pin_in = L.srcnn_host_alloc_pinned(F * w * h * 4)
        pin_out = L.srcnn_host_alloc_pinned(F * 4 * w * h * 4)
        frames = np.ctypeslib.as_array(C.cast(pin_in, C.POINTER(C.c_float)), (F, h, w))
        out = np.ctypeslib.as_array(C.cast(pin_out, C.POINTER(C.c_float)), (F, 2 * h, 2 * w))
        two = synth.frames(2, h, w, rank * F, "smooth")
        for f in range(F):
            frames[f] = two[f & 1]

        def step():
            S.check(L.srcnn_y_upscale2x_f32_stream(frames.ctypes.data, w, h, F, out.ctypes.data, 1))
        mpix_step = world * F * 4 * w * h / 1e6
        label = "%d host-resident 3840x2160 frames per rank per step, H2D + compute + D2H overlapped, hipGraph per slot" % F
    elif args.workload == "frames-graph":
        import ctypes as C
        w, h, F = 3840, 2160, args.frames
        d_in = S.DeviceBuffer(F * w * h * 4)
        d_out = S.DeviceBuffer(F * 4 * w * h * 4)
        for f in range(F):
            d_in.upload(synth.plane(h, w, synth.SEED0 + rank * F + f, "smooth"), offset=f * w * h * 4)
        st = S.Stream()
        gh = C.c_void_p()
        S.check(L.srcnn_batch_graph_create(d_in.ptr, w, h, F, d_out.ptr, st.handle, C.byref(gh)))

        def step():
            S.check(L.srcnn_batch_graph_launch(gh))
        mpix_step = world * F * 4 * w * h / 1e6
        label = "%d resident 3840x2160 frames per rank per step, replayed from one captured hipGraph" % F
    else:
        w, h, F = 1920, 1080, 64
        d_in = S.DeviceBuffer(F * w * h * 4)
        d_out = S.DeviceBuffer(F * 4 * w * h * 4)
        for f in range(F):
            d_in.upload(synth.plane(h, w, synth.SEED0 + rank * F + f, "smooth"), offset=f * w * h * 4)

        def step():
            S.check(L.srcnn_y_upscale2x_f32_batch_dev(d_in.ptr, w, h, F, d_out.ptr, None))
        mpix_step = world * F * 4 * w * h / 1e6
        label = "batch of 64 resident 1920x1080 frames per rank per step"

    for _ in range(args.warmup):
        step()
    S.sync(); barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    S.sync(); barrier()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    if dist is not None:
        import torch
        t = torch.tensor([ms], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t[0])
    if rank == 0:
        print(json.dumps({"metric": "megapixels/sec SRCNN Y-channel (2x upscale)", "value": round(mpix_step / (ms * 1e-3), 2),
                          "unit": "MPix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(ms, 4), "higher_is_better": True,
                          "scaling": "strong" if args.workload == "tiled8k" else "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": label, "mode": "strict"}}), flush=True)
    barrier()
    if args.workload == "tiled8k":
        L.srcnn_comm_destroy()
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=4, help="4K frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="frames", choices=["frames", "frames-graph", "tiled8k", "host-stream", "batch1080p"],
                    help="frames (default, the headline metric): resident 4K frames sharded across ranks; "
                         "tiled8k: ONE 7680x4320 frame -> 15360x8640, output bands across ranks + RCCL gather; "
                         "host-stream: PCIe-inclusive stream of 4K frames from host memory (hipGraph per slot); "
                         "batch1080p: 64 resident 1920x1080 frames per step; "
                         "frames-graph: the headline workload replayed from one captured hipGraph per step")
    args = ap.parse_args()
    if args.workload != "frames":
        return side_workload(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    dist = None
    if world > 1:
        # CPU-side rendezvous only (barrier + max of a scalar).  The data path has no collective, and this
        # process drives its GPU through libsrcnn_amd.so's own HIP runtime, so torch never touches the device.
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()

    import libsrcnn_amd as S
    from libsrcnn_amd import synth
    ndev = max(1, S.device_count())
    S.init(local_rank % ndev)          # identity on a real N-GPU node; lets a 1-GPU box exercise the N>1 plumbing
    S.set_mode(S.MODE_STRICT)
    L = S.lib()

    F = args.frames
    n_in, n_out = IN_W * IN_H, 4 * IN_W * IN_H
    d_in = S.DeviceBuffer(F * n_in * 4)
    d_out = S.DeviceBuffer(F * n_out * 4)
    for f in range(F):   # rank r owns frames r*F .. r*F+F-1 of the synthetic stream
        d_in.upload(synth.plane(IN_H, IN_W, synth.SEED0 + rank * F + f, "smooth"), offset=f * n_in * 4)

    def step():
        S.check(L.srcnn_y_upscale2x_f32_batch_dev(d_in.ptr, IN_W, IN_H, F, d_out.ptr, None))

    for _ in range(args.warmup):
        step()
    S.sync()

    S.profile_reset()
    S.profile_enable(True)
    barrier()
    S.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    S.sync()
    barrier()
    t1 = time.perf_counter()
    S.profile_enable(False)
    prof = S.profile_read()

    ms_per_step = (t1 - t0) * 1e3 / args.steps
    if dist is not None:
        import torch
        t = torch.tensor([ms_per_step], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_per_step = float(t[0])

    if rank == 0:
        mpix_step = world * F * n_out / 1e6
        value = mpix_step / (ms_per_step * 1e-3)
        c12_ms, c12_n = prof["conv12"]
        avg12 = c12_ms / max(c12_n, 1)
        flops12 = 2.0 * MAC_L12 * n_out                         # algorithmic FLOPs of one conv12 launch (one frame)
        achieved = flops12 / (avg12 * 1e-3) / 1e12 if avg12 > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_conv12.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        stage = {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}
        alg_bytes12 = (4 + 128) * n_out                         # layer-1+2 kernel: fp32 Y in, 32 fp32 planes out
        out = {
            "metric": "megapixels/sec SRCNN Y-channel (2x upscale)",
            "value": round(value, 2), "unit": "MPix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "stream of synthetic 3840x2160 Y frames -> 7680x4320 (2x), %d frames/GPU/step, frames "
                                   "sharded across ranks, strict (bit-exact) mode" % F,
                       "frames_per_gpu_per_step": F, "in": [IN_W, IN_H], "out": [2 * IN_W, 2 * IN_H], "mode": "strict",
                       "parallelism": "frames sharded %d-way, no data-path collective" % world},
            "roofline": {"kernel": "k_conv12_mfma (conv 9x9x1->64 + ReLU + conv 1x1x64->32 + ReLU)",
                         "bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_TFLOPS, 4), "traffic": traffic,
                         "frac_of_no_fma_ceiling": round(achieved / (PEAK_F32_TFLOPS / 2), 4),
                         "avg_launch_ms": round(avg12, 4), "launches": int(c12_n),
                         "flops_per_launch": flops12,
                         "note": "strict mode rounds product and sum separately (no FMA): ceiling is 0.5 of this peak",
                         "hbm": {"algorithmic_bytes_per_launch": alg_bytes12,
                                 "achieved_GBps": round(alg_bytes12 / (avg12 * 1e-3) / 1e9, 1) if avg12 > 0 else 0.0,
                                 "peak_GBps": PEAK_HBM_GBS,
                                 "frac": round(alg_bytes12 / (avg12 * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if avg12 > 0 else 0.0}},
            "stage_avg_ms_per_frame": stage,
            "whole_path": {"tflops": round(2.0 * MAC_ALL * mpix_step * 1e6 / (ms_per_step * 1e-3) / 1e12, 3),
                           "hbm_algorithmic_GBps": round(5 * mpix_step * 1e6 / (ms_per_step * 1e-3) / 1e9, 2)},
            "max_abs_dY_vs_cpu_ref": None,
            "device": S.device_name(),
        }
        if world == 1:
            # non-parity tiers, same frames, 2 steps each: reported beside the headline, never as `value`
            tiers = {}
            for mode, name in ((S.MODE_FAST, "fast_fp32_fma"), (S.MODE_FAST_F16, "fast_split_fp16_mfma")):
                S.set_mode(mode)
                step(); S.sync()
                t0f = time.perf_counter()
                for _ in range(2):
                    step()
                S.sync()
                tiers[name] = {"MPix/s": round(F * n_out / 1e6 / ((time.perf_counter() - t0f) / 2), 1),
                               "max_abs_dY_vs_reference": "~3e-4 (tests/test_gpu_parity.py: <= 1e-3)"}
            S.set_mode(S.MODE_STRICT)
            out["non_parity_tiers"] = tiers
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"], out["max_abs_dY_vs_cpu_ref"] = cpu_baseline(S)
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "unit": "MPix/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %s" % e}
        print(json.dumps(out), flush=True)

    barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
