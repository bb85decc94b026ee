/*
 * srcnn_amd_debug.h -- INSTRUMENTS of libsrcnn_amd.so: test hooks, diagnostics and the relaxation experiment.
 *
 * NOT part of the stable ABI (include/srcnn_amd.h, frozen at SRCNN_AMD_ABI_VERSION 5): anything here may change or go
 * between builds without a version bump.  The product's callers need none of it; the test-suite, bench.py's phase
 * table and the tools/ probes do.  A strict-only build (make STRICT_ONLY=1) keeps the symbols and answers
 * SRCNN_E_UNSUPPORTED where the instrument needs a non-parity kernel.
 */
#ifndef SRCNN_AMD_DEBUG_H
#define SRCNN_AMD_DEBUG_H

#include "srcnn_amd.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* ---- the relaxation experiment (why there is no tolerance tier) ---- */
#define SRCNN_MODE_RELAXED  3   /* the strict kernels with the roundings named by srcnn_set_relaxation() given up, layer by
                                 * layer: the instrument behind the per-layer error matrix (profiles/r04_error_matrix.txt).
                                 * NOT a parity tier: every single relaxation measures above the 1e-4 bar */
/* relaxation bits (src/libsrcnn.cpp:395-410 layer 1, :433-437 layer 2, :500-517 layer 3) */
#define SRCNN_RELAX_L1      1u  /* layer 1: acc = fma(w, y, acc) per tap (fp32 MFMA, C = acc) instead of product then add */
#define SRCNN_RELAX_L2      2u  /* layer 2: likewise over the 64 channels */
#define SRCNN_RELAX_L3_X64  4u  /* layer 3: exact products (v_fma_f64 on widened operands) instead of fp32-rounded ones;
                                 * per-channel fp64 sums and the fp32 running sum as the reference's */
#define SRCNN_RELAX_L3_F32  8u  /* layer 3: fp32 FMA chain per channel */

int         srcnn_set_relaxation(unsigned mask);   /* SRCNN_RELAX_* bits used by SRCNN_MODE_RELAXED (default L3_X64);
                                                    * returns the previous mask or <0.  Takes effect for calls that start later */

/* The table FRawScaleWeightsTable builds (src/frawscale.cpp:8-112), exposed for tests:
 * returns the window size; if left/right/weights are non-NULL fills dst_len entries
 * (weights row stride = window+1 doubles). */
int srcnn_axis_table(int filter, unsigned dst_len, unsigned src_len, int* left, int* right, double* weights);

/* Diagnostic: the fused non-parity kernel alone on an already upscaled plane (w x h), optionally with in-kernel
 * s_memtime stamps of workgroup 0 (d_dbg: 8 x 64 x 4 uint64, or NULL).  Used by tools/fused_timeline.py. */
int srcnn_fused_diag(const float* d_up, unsigned w, unsigned h, float* d_out, unsigned long long* d_dbg, void* stream);

/* Test hook: number of contribution tables currently cached (bounded, only unreferenced tables are evicted) and
 * of ProcessSRCNN lanes created so far (at most 4 per context), both summed over the contexts. */
int srcnn_debug_counts(int* tables, int* lanes);

/* Every SRCNN_* environment switch of the library is read ONCE, when the library is loaded, into one table
 * (libsrcnn_amd/csrc/srcnn_settings.hpp): this prints what the process runs with, one "NAME=value (default D)  -- effect" line
 * per switch (markdown != 0: the table rows of DESIGN.md section 6).  The text is written to buf (NUL-terminated, truncated to
 * cap; buf may be NULL); the return value is the length the whole text needs.  No device needed. */
int srcnn_debug_settings(char* buf, size_t cap, int markdown);

/* Diagnostic: the shader clock of every layer-1+2 launch.  While on, each launch records the shader-cycle and 100 MHz counters
 * over the lifetime of its first workgroup (cycles / ticks x 100 = MHz; ticks / 100 = microseconds).  probe(on) resets the
 * record and returns the previous setting; read() synchronises the device, returns the number of launches recorded on
 * `context` (at most 8192 are kept) and writes up to `cap` of them in launch order. */
int srcnn_debug_clock_probe(int on);
int srcnn_debug_clock_read(int context, unsigned long long* cycles, unsigned long long* ticks, int cap);

/* Test hook (no device needed): the band cut points srcnn_process_u8 uses for output rows [r0, r1) of a dw-wide image under
 * the current workspace limit; returns their number (first = r0, last = r1), writes at most `cap` of them. */
int srcnn_debug_band_plan(unsigned r0, unsigned r1, unsigned dw, int one_of_many, unsigned* cuts, int cap);

/* Diagnostic: where the time of the calling thread's last large (banded) srcnn_process_u8 / ProcessSRCNN went -- what
 * SRCNN_TRACE prints, as numbers.  ms[0..4]: milliseconds since the call's share began at which (0) setup was done (lane,
 * tables, scratch, staging, helper threads), (1) the first band was staged in and its kernels queued, (2) the last band's
 * kernels had finished, (3) the last band's D2H had landed, (4) the last band had been copied out to the caller's buffer;
 * ms[5]: the number of bands.  Returns how many values exist (0: no banded call on this thread yet), writes at most cap. */
int srcnn_debug_process_phases(double* ms, int cap);

/* Diagnostic: how the process's last srcnn_y_upscale2x_f32_stream call launched its frames, summed over the contexts:
 * frames replayed from a hipGraph, frames launched plainly, and whether use_graph = 1 retired its graphs because replay
 * burnt host CPU (SRCNN_GRAPH_MAX_CPU_PCT).  Any pointer may be NULL. */
int srcnn_debug_stream_mode(unsigned* graph_frames, unsigned* plain_frames, int* fell_back);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* SRCNN_AMD_DEBUG_H */
