// srcnn_settings.hpp -- every SRCNN_* environment switch of the library, in ONE table.
//
// The switches are read once, when the library is loaded (Settings::from_env() in the Global constructor), and never again:
// what a process runs with is what srcnn_debug_settings() prints, no launcher keeps a private `static getenv`, and the table
// below is the single source of DESIGN.md section 6 (tools/gen_settings_table.py prints it; tests/test_abi.py checks that
// the document carries exactly that text).  Nothing that must agree ACROSS ranks may depend on a switch: the gather table of
// the tiled multi-GPU frame is a function of (width, height, ranks, pieces) alone (srcnn::tiled_cuts).
//
// X(kind, member, "ENV_NAME", default, "values", "effect")      kind: B = bool (0/1), I = integer, S = string
#pragma once
#include <cstddef>
#include <string>

#define SRCNN_SETTINGS(X)                                                                                                          \
    /* ---- host side ---- */                                                                                                        \
    X(S, devices,          "SRCNN_DEVICES",          "",    "unset = device 0, `all`, `0,1,...`", "contexts of a process that never calls `srcnn_init*` (an id may repeat: virtual contexts on one device)") \
    X(I, max_workspace_mb, "SRCNN_MAX_WORKSPACE_MB", 4608, "MiB (API twin `srcnn_set_workspace_limit`)", "layer-2 scratch one pass may hold; larger ranges are banded, bit-identically.  The default holds a whole 7680x4320 output frame (4.25 GB); lower it where memory is short: 2048 costs -1...+2 % by box, 512 +2 %, 256 +9 % (`profiles/r06_lowmem.txt`); 16384 until round 6") \
    X(I, max_lanes,        "SRCNN_MAX_LANES",        4,     "1...64", "concurrent `ProcessSRCNN` calls per context before callers queue") \
    X(B, numa,             "SRCNN_NUMA",             1,     "0/1", "place page-locked staging on the device's NUMA node (the caller's memory policy is saved and restored)") \
    X(I, comm_timeout_ms,  "SRCNN_COMM_TIMEOUT_MS",  60000, "ms, 0 = none (API twin `srcnn_comm_set_timeout_ms`)", "deadline of every wait on a peer") \
    X(B, comm_check,       "SRCNN_COMM_CHECK",       1,     "0/1", "all-reduce a checksum of every new gather table across the ranks before the first gather that uses it") \
    X(S, rccl_lib,         "SRCNN_RCCL_LIB",         "",    "path", "load this RCCL instead of the one next to the bound `libamdhip64` (site builds; the test-suite's one-device stand-in).  A trust boundary: the named file's code runs in the process; refused under secure execution (`AT_SECURE`)") \
    X(I, async_chain,      "SRCNN_ASYNC_CHAIN",      1,     "0/1/2", "order the kernels of consecutive asynchronous `ProcessSRCNN` jobs: 0 = not at all, 1 = on the host, 2 = with device-side event waits") \
    X(B, trace,            "SRCNN_TRACE",            0,     "0/1", "one stderr line per `srcnn_process_u8` share (setup / bands / stamps)") \
    X(B, roctx,            "SRCNN_ROCTX",            0,     "0/1", "roctx ranges for `rocprofv3 --marker-trace`") \
    X(S, bands,            "SRCNN_BANDS",            "",    "`f0,f1,...`", "band fractions of a large `ProcessSRCNN` image instead of the planned cuts") \
    X(I, graph_max_cpu_pct, "SRCNN_GRAPH_MAX_CPU_PCT", 10,  "0...100, 0 = never fall back", "`srcnn_y_upscale2x_f32_stream(use_graph = 1)`: hipGraph replay is kept only while the process's CPU time per replayed frame stays below this share of the frame's wall time (plain launches cost 6-8 % of a 4K frame; replay on ROCm 7.2 costs 20 % in a fresh process and ~100 % -- a runtime thread spinning from launch to completion -- in a process with more streams); above it the stream goes on with plain launches: same kernels, same overlap") \
    X(B, prefault,         "SRCNN_PREFAULT",         1,     "0/1", "pre-fault the fresh result pages of `ProcessSRCNN` in the background") \
    X(I, prefault_threads, "SRCNN_PREFAULT_THREADS", 1,     ">= 1", "helpers of that pre-faulting") \
    X(B, thp,              "SRCNN_THP",              1,     "0/1", "huge-page hint on large result blocks") \
    X(B, spin_wait,        "SRCNN_SPIN_WAIT",        0,     "0/1", "host waits through the runtime (holds a core) instead of polling") \
    X(B, device_wait_in,   "SRCNN_DEVICE_WAIT_IN",   0,     "0/1", "bands wait for their stage-in device-side (a runtime thread stays busy)") \
    /* ---- kernel selection: every choice is bit-identical and tested (production + one fallback per kernel) ---- */               \
    X(B, conv12_queue,     "SRCNN_CONV12_QUEUE",     1,     "0/1", "layer 1+2: tiles drawn from a global counter vs dealt with a static stride") \
    X(B, conv12_dma,       "SRCNN_CONV12_DMA",       1,     "0/1", "layer 1+2: weights and Y tiles staged by LDS-DMA one tile ahead vs load, wait, `ds_write`") \
    X(B, conv12_spread,    "SRCNN_CONV12_SPREAD",    1,     "0/1", "layer 1+2: small launches dealt in quarter tiles over up to 4x more workgroups") \
    X(B, conv3_wdma,       "SRCNN_CONV3_WDMA",       1,     "0/1", "layer 3: packed weight image staged by LDS-DMA vs a plain loop") \
    X(B, conv3_off64,      "SRCNN_CONV3_OFF64",      0,     "0/1", "layer 3: 64-bit plane offsets (chosen automatically for planes of 4 GiB or more)") \
    X(B, rs_dma,           "SRCNN_RS_DMA",           1,     "0/1", "resampler: `k_rs2d_dma` (source patch by LDS-DMA) vs `k_rs2d` for plane sources") \
    X(I, rs_tpb,           "SRCNN_RS_TPB",           0,     "0 = auto, 1...16", "resampler: row tiles a block marches through") \
    X(B, resample_2pass,   "SRCNN_RESAMPLE_2PASS",   0,     "0/1", "resampler: always the two generic passes (`k_resample_cols/rows`, what down-scales use); implies the unfused colour shell") \
    X(B, shell_unfused,    "SRCNN_SHELL_UNFUSED",    0,     "0/1", "colour shell as split + plane resamples + merge instead of fused into the resampler")

namespace srcnn {

struct Settings {
#define SRCNN_SET_B(m, d) bool m = (d) != 0;
#define SRCNN_SET_I(m, d) long m = (d);
#define SRCNN_SET_S(m, d) std::string m = d;
#define X(kind, member, env, def, values, effect) SRCNN_SET_##kind(member, def)
    SRCNN_SETTINGS(X)
#undef X
#undef SRCNN_SET_B
#undef SRCNN_SET_I
#undef SRCNN_SET_S
    static Settings from_env();
    // one line per switch: "NAME=value (default D)  -- effect", or (markdown != 0) the rows of DESIGN.md section 6
    std::string describe(bool markdown) const;
};

const Settings& settings();     // the process's settings, read when the library was loaded

}  // namespace srcnn
