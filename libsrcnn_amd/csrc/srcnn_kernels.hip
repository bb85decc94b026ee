// srcnn_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the SRCNN Y path.
//
// Reference behaviour reproduced (rageworx/libsrcnn, paths relative to its tree):
//   k_rs2d_dma / k_rs2d   FRAWResizeEngine::scale for up-scales, both passes in one kernel  src/frawscale.cpp:238-385
//   k_resample_cols/rows  verticalFilter / horizontalFilter, generic (down-scales, long tables)   src/frawscale.cpp:288-385
//   k_conv12_mfma         64 x convolution99 + 32 x convolution11                src/libsrcnn.cpp:350-447
//   k_conv3               convolution55                                          src/libsrcnn.cpp:449-529
// (round 5: one production form and one fallback per kernel -- the A/B geometries of rounds 1-4, the VALU-only layer-1+2
//  kernel, the round-2 resamplers and the unfused fp16 tier are gone; their measurements stay in docs/HISTORY.md)
//   k_conv1/2_planes      the two layers unfused, 64 / 32 planes in HBM (stage-level parity entry points)
//   k_rgb_split / k_ycc_merge  colour shell                                      src/libsrcnn.cpp:233-308,889-905
//
// STRICT kernels keep the reference's evaluation order and roundings exactly: one rounded fp32 product then
// one rounded fp32 add per tap (no FMA; this TU is built with -ffp-contract=off and the pragma below), taps in
// the reference's loop order, fp64 where the reference uses double.  The non-parity tiers are template
// parameters / separate kernels and are never selected unless srcnn_set_mode asks for them; -DSRCNN_STRICT_ONLY
// (make STRICT_ONLY=1) compiles no instance of them at all.
//
// Data layout in HBM: every image is planar float32, row-major; activation stacks are [channel][row][col]
// with a caller-given plane stride.  Weights live in __constant__ memory, re-laid out once on the host:
//   w1t[tap][k]   (tap = 9*row_off + col_off)   <- weights_conv1_data[k][row_off][col_off]
//   w2 [m][f]                                    <- weights_conv2_data[m][f]
//   w3 [m][dy][dx]                               <- weights_conv3_data[m][dx][dy]   (transposed!)
// VALU kernels read them as SGPR operands (wave-uniform addresses -> s_load); the MFMA kernels re-stage
// them into LDS per block in operand (lane) order.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "srcnn_kernels.h"
#include "srcnn_settings.hpp"

#pragma clang fp contract(off)

namespace srcnn {

__constant__ DevWeights cW;

hipError_t upload_weights(const DevWeights& w)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(cW), &w, sizeof(DevWeights));
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// =============================================================================================
// Resampler: table-driven separable passes, fp64 multiply then fp64 add, fp32 store.
// One thread per output sample; consecutive lanes = consecutive x (coalesced).
// Rows are addressed in "global" row numbers with a base row per buffer so the same kernels
// serve whole frames and horizontal bands.
// =============================================================================================
__global__ __launch_bounds__(256) void k_resample_cols(   // vertical pass: [src_h x w] -> rows [r0,r1) of [dst_h x w]
    const float* __restrict__ src, int w, int src_row_base,
    float* __restrict__ dst, int dst_row0, int dst_rows,
    const int* __restrict__ first, const int* __restrict__ taps, const double* __restrict__ wt, int stride)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    for (int ry = blockIdx.y; ry < dst_rows; ry += gridDim.y) {     // gridDim.y is capped at 65535
        const int y = dst_row0 + ry;
        const int s0 = first[y], n = taps[y];
        const double* wr = wt + (size_t)y * stride;
        double acc = 0.0;
        for (int t = 0; t < n; ++t) {
            const double px = (double)src[(size_t)(s0 + t - src_row_base) * w + x];
            acc = acc + wr[t] * px;
        }
        dst[(size_t)ry * w + x] = (float)acc;
    }
}

__global__ __launch_bounds__(256) void k_resample_rows(   // horizontal pass: [rows x src_w] -> [rows x dst_w]
    const float* __restrict__ src, int src_w, float* __restrict__ dst, int dst_w, int rows,
    const int* __restrict__ first, const int* __restrict__ taps, const double* __restrict__ wt, int stride)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= dst_w) return;
    const int s0 = first[x], n = taps[x];
    const double* wr = wt + (size_t)x * stride;
    for (int y = blockIdx.y; y < rows; y += gridDim.y) {
        const float* in = src + (size_t)y * src_w + s0;
        double acc = 0.0;
        for (int t = 0; t < n; ++t) acc = acc + wr[t] * (double)in[t];
        dst[(size_t)y * dst_w + x] = (float)acc;
    }
}

// =============================================================================================
// Round-3 resampler family k_rs2d: both passes of an up-scale in ONE kernel (vertical first, then horizontal, as
// src/frawscale.cpp:238-278 orders them), same operations in the same order as k_resample_cols + k_resample_rows.
//   * a block owns 256 output columns and marches through `tpb` tiles of 16 rows; a lane owns FOUR adjacent output
//     columns (16-byte stores, one wave = one 1 KB row segment) and keeps their horizontal weights and LDS tap addresses
//     in registers for the whole march (the weight table, not the image, was the dominant traffic of the round-2 kernel);
//   * a tile's source span is known in closed form: the tables are monotone (checked on the host when they are built),
//     so first[] of the tile's first row/column and first[]+taps[] of its last give the patch -- a handful of scalar
//     loads, no reduction -- and the patch is fetched in one batch of independent loads;
//   * no branch inside either tap loop.  Vertical: a wave takes one tile row at a time, so the tap count is wave-uniform
//     and selects a fully unrolled body (LDS row stride LW is a template parameter: every tap is an immediate offset).
//     Horizontal: a lane's four columns may have different tap counts; the missing taps get weight 0.0 and read the zero
//     column every LDS row ends with, which leaves the fp64 accumulator bit-for-bit unchanged (see the kernel);
//   * the source can be an interleaved 8-bit RGB(A) image (KIND 1: Y computed per sample exactly like k_rgb_split), and
//     the sink can be the colour merge (KIND 2: the block resamples Cb, Cr (and A) from the source image and merges them
//     with the finished Y' rows straight into interleaved u8 -- src/libsrcnn.cpp:274-308 -- so the destination-size chroma
//     planes never exist).
// LDS (dynamic): vertical weights, then per plane the source patch [sr][LW] and the intermediate rows [16][LW] (fp32,
// i.e. rounded after the vertical pass like the reference's intermediate image).  sr is the largest row span any tile of
// THIS launch has, computed on the host from the table's host copy.  Measured (8K plane, 166 MB): 0.075 ms = 2.2 TB/s
// (round 2: 0.153 ms).  44 VALU instructions per output sample (3 of 5 of them fp64) keep the SIMDs busy 55 % of the time
// at 4 waves per SIMD (110 VGPRs); the rest is exposed latency around the two barriers of a tile.  Prefetching the next
// tile's patch into registers while the current tile computes was tried and is SLOWER (0.088 ms: the 9 extra values per
// plane push the kernel to 168 VGPRs = 3 waves per SIMD; capped at 128 VGPRs it spills, 0.091 ms) -- occupancy hides the
// latency better than software pipelining does here.  Nor does more occupancy help: with TWO columns per lane (80 VGPRs,
// 6 waves per SIMD) the kernel takes 0.079-0.088 ms.  The tap itself (v_cvt_f64_f32 + v_mul_f64 + v_add_f64) issues at
// 4.9 cycles per instruction (profiles/r03_valu_rates.txt; v_fma_f64(w, x, 0) instead of v_mul_f64 is no faster), which
// puts the pure-issue floor of the 8.6 taps per output sample at ~0.042 ms.  A 244-column tile (source span 127: two full
// 64-lane sweeps of the vertical pass instead of two and a 6-lane third) does not help either: 0.076 ms.
// =============================================================================================
struct Rs2dArgs {
    const float* src_plane; const unsigned char* src_rgb; int src_w;
    float* dst;                               // KIND 0/1: rows [dst_row0, +dst_rows), row dst_row0 at offset 0
    const float* yp; unsigned char* out_rgb; unsigned char* out_conv;   // KIND 2 (same row convention)
    int dst_w, dst_row0, dst_rows;
    int sr, tpb;                              // patch rows in LDS; row tiles a block marches through
    const int* vfirst; const int* vtaps; const double* vwt; int vstride;
    const int* hfirst; const int* htaps; const double* hwt; int hstride;
    int vec;                                  // 16-byte / dword accesses are aligned for this launch
};

__device__ __forceinline__ unsigned char to_u8_sat(float v)
{   // MIN(255.f, v) then MAX(0.f, .) then truncating cast, in the reference's macro forms
    v = (255.f < v) ? 255.f : v;
    v = (0.f > v) ? 0.f : v;
    return (unsigned char)v;
}

template <int KIND, int D>
__device__ __forceinline__ void rs_load(const Rs2dArgs& a, int r, int c, float* v)
{
    if constexpr (KIND == 0) {
        v[0] = a.src_plane[(size_t)r * a.src_w + c];
    } else {
        const unsigned char* q = a.src_rgb + ((size_t)r * a.src_w + c) * D;
        const float R = (float)q[0], Gc = (float)q[1], B = (float)q[2];
        if constexpr (KIND == 1) {
            v[0] = (0.299f * R) + (0.587f * Gc) + (0.114f * B);                  // src/libsrcnn.cpp:251-256
        } else {
            v[0] = 128.f - (0.1687f * R) - (0.3313f * Gc) + (0.5f * B);
            v[1] = 128.f + (0.5f * R) - (0.4187f * Gc) - (0.0813f * B);
            if constexpr (D == 4) v[2] = (float)q[3];
        }
    }
}

constexpr int RS_TH = 16;                      // tile rows
constexpr int RS_MAX_TPB = 16;                 // row tiles a block may march through (its horizontal weights are loaded once)
constexpr int rs_lds_bytes(int np, int maxt, int sr, int lw)
{
    // vertical weights / first rows / tap counts and the source patch are double-buffered; the intermediate rows are not
    return 2 * (RS_TH * maxt * 8 + RS_TH * 2 * 4) + np * (2 * sr + RS_TH) * lw * 4;
}

// One tile row of the vertical pass with a compile-time tap count: source rows rb .. rb+CNT-1 of the patch are LW floats
// apart, so every ds_read after the first address is an immediate offset.  Same products, same order as k_resample_cols.
template <int CNT, int NP, int MAXT, int LW, class MID = float>
__device__ __forceinline__ void rs_vertical_row(const float* raw, int raw_plane, MID* midrow, const double* vwr, int rb, int cn, int lane)
{
    double wg[CNT];
#pragma unroll
    for (int t = 0; t < CNT; ++t) wg[t] = vwr[t];
    for (int c = lane; c < cn; c += 64) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float* col = raw + p * raw_plane + rb * LW + c;
            float xs[CNT];
#pragma unroll
            for (int t = 0; t < CNT; ++t) xs[t] = col[t * LW];
            double acc = 0.0;
#pragma unroll
            for (int t = 0; t < CNT; ++t) acc = acc + wg[t] * (double)xs[t];
            midrow[p * RS_TH * LW + c] = (MID)(float)acc;       // rounded to fp32 like the reference's intermediate image
        }
    }
}

template <int KIND, int D, int MAXT, int LW, bool CONV>
__global__ __launch_bounds__(256) void k_rs2d(const Rs2dArgs a)
{
    constexpr int TH = RS_TH;
    constexpr int NP = (KIND == 2) ? D - 1 : 1;
    constexpr int U = 4;                       // patch items per thread per batch of independent loads
    constexpr int RPT = TH / 4;                // rows per thread in the horizontal pass
    constexpr int ZC = LW - 1;                 // the zero column every mid row ends with
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds[];
    double* vw2 = reinterpret_cast<double*>(rs_lds);           // [2][TH][MAXT]
    int* vf2 = reinterpret_cast<int*>(vw2 + 2 * TH * MAXT);    // [2][TH] first source row, relative to the patch
    int* vn2 = vf2 + 2 * TH;                                   // [2][TH] taps
    float* mid = reinterpret_cast<float*>(vn2 + 2 * TH);       // [NP][TH][LW]   (rows wv, wv+4, ... belong to wave wv alone)
    float* raw2 = mid + NP * TH * LW;                          // [2][NP][sr][LW]
    const int raw_plane = a.sr * LW;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x0t = blockIdx.x * 256;
    const int xl = min(x0t + 256, a.dst_w) - 1;
    // closed-form column span (monotone tables): wave-uniform scalar loads
    const int c0 = a.hfirst[x0t], cn = a.hfirst[xl] + a.htaps[xl] - c0;

    // ---- once per block: the horizontal weights of this thread's four columns.  Taps beyond a column's count are made
    //      harmless WITHOUT a branch: weight 0.0 and a source address that points at the zero column every mid row ends
    //      with (0.0 * 0.0 = +0.0, and acc + 0.0 is acc: the accumulator starts at +0.0 and can never become -0.0).  Reading
    //      a real sample with weight 0 would not do: 0 * inf is NaN, and the reference never touches those samples
    //      (trailing zero taps are trimmed, src/frawscale.cpp:95-107). ----
    const float* hp[4][MAXT];                  // LDS address of tap t of column j in mid row `wv` of plane 0
    double w[4][MAXT];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xc = min(x0t + 4 * lane + j, a.dst_w - 1);
        const int s0 = a.hfirst[xc] - c0, n = a.htaps[xc];
        const double* wr = a.hwt + (size_t)xc * a.hstride;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            w[j][t] = t < n ? wr[t] : 0.0;
            hp[j][t] = mid + wv * LW + (t < n ? s0 + t : ZC);
        }
    }
    if (tid < NP * TH) mid[tid * LW + ZC] = 0.f;                 // the zero column (never overwritten: cn < LW)

    const int x = x0t + 4 * lane;
    // Tile loop with ONE workgroup barrier per tile.  The patch and the vertical weights are double-buffered, and a wave
    // consumes in the horizontal pass exactly the intermediate rows it produced in the vertical pass (rows wv, wv+4, ...), so
    // the hand-over between the passes is wave-local.  A wave that runs ahead may already fetch tile s+1 into the other
    // buffer while slower waves still read tile s; it cannot get two tiles ahead, because the barrier of tile s+1 waits for
    // every wave to have left tile s.
    for (int sub = 0; sub < a.tpb; ++sub) {
        const int ry0 = (blockIdx.y * a.tpb + sub) * TH;
        if (ry0 >= a.dst_rows) break;
        const int rows = min(TH, a.dst_rows - ry0);
        const int y0 = a.dst_row0 + ry0, yl = y0 + rows - 1;
        const int vmin = a.vfirst[y0], nsrc = a.vfirst[yl] + a.vtaps[yl] - vmin;     // closed-form row span
        double* vw = vw2 + (sub & 1) * TH * MAXT;
        int* vf = vf2 + (sub & 1) * TH;
        int* vn = vn2 + (sub & 1) * TH;
        float* raw = raw2 + (sub & 1) * NP * raw_plane;

        // ---- everything this tile needs from global memory is requested now ----
        // (1) KIND 2: the Y' values this thread will merge with
        float yv[RPT][4];
        if constexpr (KIND == 2) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int r = wv + 4 * k;
#pragma unroll
                for (int j = 0; j < 4; ++j) yv[k][j] = 0.f;
                if (r < rows && x < a.dst_w) {
                    const float* yr = a.yp + (size_t)(ry0 + r) * a.dst_w + x;
                    if (a.vec) { const float4 q = *reinterpret_cast<const float4*>(yr); yv[k][0] = q.x; yv[k][1] = q.y; yv[k][2] = q.z; yv[k][3] = q.w; }
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (x + j < a.dst_w) yv[k][j] = yr[j];
                    }
                }
            }
        }
        // (2) the vertical weights of the tile's rows
        if (tid < TH * MAXT) {
            const int r = tid / MAXT, t = tid - r * MAXT;
            if (r < rows) {
                const int y = y0 + r;
                const int cnt = a.vtaps[y];
                vw[tid] = t < cnt ? a.vwt[(size_t)y * a.vstride + t] : 0.0;
                if (t == 0) { vf[r] = a.vfirst[y] - vmin; vn[r] = cnt; }
            }
        }
        // (3) the source patch: rows [vmin, vmin+nsrc) x columns [c0, c0+cn), item i = r*cn + c, U independent loads at a time
        {
            const int total = nsrc * cn;
            const int qstep = 256 / cn, rstep = 256 - qstep * cn;      // (r, c) advance of 256 items
            int r = tid / cn, c = tid - r * cn;
            for (int base = 0; base < total; base += 256 * U) {
                float v[U][NP];
                int ro[U], co[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    ro[u] = r; co[u] = c;
                    if (base + tid + 256 * u < total) rs_load<KIND, D>(a, vmin + r, c0 + c, v[u]);
                    c += rstep; r += qstep;
                    if (c >= cn) { c -= cn; ++r; }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (base + tid + 256 * u < total) {
#pragma unroll
                        for (int p = 0; p < NP; ++p) raw[p * raw_plane + ro[u] * LW + co[u]] = v[u][p];
                    }
            }
        }
        __syncthreads();
        // ---- vertical pass: a wave takes one tile row at a time (first source row, tap count and weights are wave-uniform),
        //      lanes run over the source columns; the tap count selects a fully unrolled body ----
        for (int r = wv; r < rows; r += 4) {
            const int rb = __builtin_amdgcn_readfirstlane(vf[r]), cnt = __builtin_amdgcn_readfirstlane(vn[r]);
            float* mrow = mid + r * LW;
            const double* vwr = vw + r * MAXT;
            switch (cnt) {
#define RS_V(N) case N: if constexpr (N <= MAXT) rs_vertical_row<N, NP, MAXT, LW>(raw, raw_plane, mrow, vwr, rb, cn, lane); break;
            RS_V(1) RS_V(2) RS_V(3) RS_V(4) RS_V(5) RS_V(6) RS_V(7) RS_V(8)
#undef RS_V
            default: break;
            }
        }
        // this wave's intermediate rows are complete and visible to its own lanes: no workgroup barrier needed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- horizontal pass (same taps, same order as k_resample_rows), four columns per lane, then the sink ----
        if (x < a.dst_w) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int r = wv + 4 * k;
                if (r >= rows) break;
                float o[NP][4];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float xs[MAXT];
#pragma unroll
                        for (int t = 0; t < MAXT; ++t) xs[t] = hp[j][t][(p * TH + 4 * k) * LW];      // immediate offset
                        double acc = 0.0;
#pragma unroll
                        for (int t = 0; t < MAXT; ++t) acc = acc + w[j][t] * (double)xs[t];
                        o[p][j] = (float)acc;
                    }
                }
                const size_t pix = (size_t)(ry0 + r) * a.dst_w + x;
                if constexpr (KIND != 2) {
                    float* dr = a.dst + pix;
                    if (a.vec) *reinterpret_cast<float4*>(dr) = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (x + j < a.dst_w) dr[j] = o[0][j];
                    }
                } else {
                    unsigned char px[4][4];
                    unsigned cw = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float fy = yv[k][j], cb = o[0][j] - 128.f, cr = o[1][j] - 128.f;     // src/libsrcnn.cpp:289-299
                        px[j][0] = to_u8_sat(fy + 45.f * cr / 32.f);
                        px[j][1] = to_u8_sat(fy - (11.f * cb + 23.f * cr) / 32.f);
                        px[j][2] = to_u8_sat(fy + 113.f * cb / 64.f);
                        px[j][3] = 0;
                        if constexpr (D == 4) px[j][3] = to_u8_sat(o[NP - 1][j]);
                        cw |= (unsigned)(unsigned char)fy << (8 * j);                              // src/libsrcnn.cpp:897-901
                    }
                    unsigned char* orow = a.out_rgb + pix * D;
                    if (a.vec) {
                        unsigned wds[D] = {};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int ch = 0; ch < D; ++ch) {
                                const int byte = j * D + ch;
                                wds[byte >> 2] |= (unsigned)px[j][ch] << (8 * (byte & 3));
                            }
                        unsigned* o32 = reinterpret_cast<unsigned*>(orow);
#pragma unroll
                        for (int q = 0; q < D; ++q) o32[q] = wds[q];
                        if constexpr (CONV) *reinterpret_cast<unsigned*>(a.out_conv + pix) = cw;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (x + j < a.dst_w) {
#pragma unroll
                                for (int ch = 0; ch < D; ++ch) orow[j * D + ch] = px[j][ch];
                                if constexpr (CONV) a.out_conv[pix + j] = (unsigned char)(cw >> (8 * j));
                            }
                    }
                }
            }
        }
        // the wave's own next vertical pass overwrites its intermediate rows: its lanes must be done reading them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---------------------------------------------------------------------------------------------
// k_rs2d_dma: the plane -> plane member of the family with the source patch moved by LDS-DMA (global_load_lds_dword: no
// VGPR destination), ONE TILE AHEAD of the arithmetic.  k_rs2d asks for a tile's patch at the top of the tile and then
// waits for it; holding the next patch in registers instead costs a wave per SIMD (see above).  The DMA needs neither:
// while tile s is computed, tile s+1's patch lands in the other LDS buffer.  A wave moves patch rows wv, wv+4, ...; one
// wave-instruction carries 64 consecutive source columns of one row (the LDS destination of a DMA is wave-uniform base +
// lane * 4, which is exactly a row segment of the [row][LW] patch).  The DMA is an asm statement (hipcc would otherwise
// wait for it before the first ds_read that follows); its completion is awaited explicitly before the tile's barrier.
// Same arithmetic, same order as k_rs2d<0>: bit-identical.  Measured on an 8K plane: 0.056 ms = 3.0 TB/s (k_rs2d<0>:
// 0.072 ms); the steps are listed in DESIGN.md 4.3.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void rs_dma_dword(const float* gsrc, const float* lds_dst)
{
    unsigned keep;
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// The same with "uniform base (SGPR pair) + per-lane 32-bit byte offset" addressing: no 64-bit address arithmetic on the VALU.
__device__ __forceinline__ void rs_dma_dword_s(const void* sbase, unsigned voff, const float* lds_dst)
{
    unsigned keep;
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)lds_dst;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}

// wave-uniform table entries through the scalar cache (the tables are written long before any launch that reads them)
__device__ __forceinline__ int rs_sload(const int* p, int i)
{
    return reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(p))[i];
}

template <int MAXT, int LW>
__global__ __launch_bounds__(256) void k_rs2d_dma(const Rs2dArgs a)
{
    constexpr int TH = RS_TH;
    constexpr int RPT = TH / 4;
    constexpr int ZC = LW - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds[];
    double* vw2 = reinterpret_cast<double*>(rs_lds);           // [2][TH][MAXT]
    int* vf2 = reinterpret_cast<int*>(vw2 + 2 * TH * MAXT);    // [2][TH]
    int* vn2 = vf2 + 2 * TH;                                   // [2][TH]
    double* mid = reinterpret_cast<double*>(vn2 + 2 * TH);     // [TH][LW]  fp32-rounded values, kept as doubles (see below)
    float* raw2 = reinterpret_cast<float*>(mid + TH * LW);     // [2][sr][LW]
    const int raw_plane = a.sr * LW;

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0t = blockIdx.x * 256;
    const int xl = min(x0t + 256, a.dst_w) - 1;
    const int c0 = rs_sload(a.hfirst, x0t), cn = rs_sload(a.hfirst, xl) + rs_sload(a.htaps, xl) - c0;
    const int chunks = (cn + 63) >> 6;

    // this thread's slot of a tile's vertical weight table, fetched one tile ahead like the patch
    const int wr_r = tid / MAXT, wr_t = tid - wr_r * MAXT;
    double wreg = 0.0;
    int freg = 0, nreg = 0;
    auto fetch = [&](int sub) {
        const int ry0 = (blockIdx.y * a.tpb + sub) * TH;
        const int rows = min(TH, a.dst_rows - ry0);
        const int y0 = a.dst_row0 + ry0, yl = y0 + rows - 1;
        const int vmin = rs_sload(a.vfirst, y0), nsrc = rs_sload(a.vfirst, yl) + rs_sload(a.vtaps, yl) - vmin;
        if (tid < TH * MAXT && wr_r < rows) {                  // three independent loads; the tap-count select happens at the use
            const int y = y0 + wr_r;
            nreg = a.vtaps[y];
            freg = a.vfirst[y] - vmin;
            wreg = a.vwt[(size_t)y * a.vstride + min(wr_t, a.vstride - 1)];
        }
        float* raw = raw2 + (sub & 1) * raw_plane;
        const float* g = a.src_plane + (size_t)vmin * a.src_w + c0 + lane;
        for (int r = wv; r < nsrc; r += 4)
            for (int ch = 0; ch < chunks; ++ch)
                if (ch * 64 + lane < cn) rs_dma_dword(g + (size_t)r * a.src_w + ch * 64, raw + r * LW + ch * 64);
    };

    fetch(0);              // tile 0's patch is on its way while the horizontal weights below are fetched


    const double* hp[4][MAXT];
    double w[4][MAXT];
    int nmax = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // a lane owns columns 2*lane, 2*lane+1 of each 128-column half of the tile: neighbouring lanes then read
        // neighbouring intermediate samples (for a 2x up-scale; four adjacent columns per lane put the lanes two dwords apart: a
        // two-way LDS bank conflict on every horizontal tap)
        const int xc = min(x0t + 128 * (j >> 1) + 2 * lane + (j & 1), a.dst_w - 1);
        const int s0 = a.hfirst[xc] - c0, n = a.htaps[xc];
        const double* wr = a.hwt + (size_t)xc * a.hstride;
        nmax = max(nmax, n);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            w[j][t] = t < n ? wr[t] : 0.0;
            hp[j][t] = mid + wv * LW + (t < n ? s0 + t : ZC);
        }
    }
    if (tid < TH) mid[tid * LW + ZC] = 0.0;
    // The table's widest column sets MAXT, but it is usually an edge case (2x Mitchell: 4 taps everywhere, 5 on two columns
    // of 7680).  A wave none of whose columns uses the last tap skips it -- that tap would only add +0.0.
    const bool short_taps = MAXT > 1 && __builtin_amdgcn_ballot_w64(nmax > MAXT - 1) == 0;

    const int xa = x0t + 2 * lane, xb = xa + 128;
    for (int sub = 0; sub < a.tpb; ++sub) {
        const int ry0 = (blockIdx.y * a.tpb + sub) * TH;
        if (ry0 >= a.dst_rows) break;
        const int rows = min(TH, a.dst_rows - ry0);
        double* vw = vw2 + (sub & 1) * TH * MAXT;
        int* vf = vf2 + (sub & 1) * TH;
        int* vn = vn2 + (sub & 1) * TH;
        const float* raw = raw2 + (sub & 1) * raw_plane;
        if (tid < TH * MAXT && wr_r < rows) {
            vw[tid] = wr_t < nreg ? wreg : 0.0;
            if (wr_t == 0) { vf[wr_r] = freg; vn[wr_r] = nreg; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of the tile's patch has landed
        __syncthreads();                                        // ... and so has everybody else's; tile sub-1 is finished everywhere
        if (sub + 1 < a.tpb && ry0 + TH < a.dst_rows) fetch(sub + 1);

        // vertical pass.  Full 64-column sweeps row by row (wave-uniform tap count -> unrolled body); what is left of a row
        // (7 of 135 columns for a 2x up-scale: a sweep with 7 lanes at work) is done for the wave's four rows in ONE sweep,
        // 16 lanes per row, with per-lane weights and a select instead of the wave-uniform tap count.
        const int full = cn & ~63, rem = cn - full;
        const bool tail = rem > 0 && rem <= 16;
        const int cv = tail ? full : cn;
        for (int r = wv; r < rows; r += 4) {
            const int rb = __builtin_amdgcn_readfirstlane(vf[r]), cnt = __builtin_amdgcn_readfirstlane(vn[r]);
            double* mrow = mid + r * LW;
            const double* vwr = vw + r * MAXT;
            switch (cnt) {
#define RS_V(N) case N: if constexpr (N <= MAXT) rs_vertical_row<N, 1, MAXT, LW, double>(raw, raw_plane, mrow, vwr, rb, cv, lane); break;
            RS_V(1) RS_V(2) RS_V(3) RS_V(4) RS_V(5) RS_V(6) RS_V(7) RS_V(8)
#undef RS_V
            default: break;
            }
        }
        if (tail) {
            const int r = wv + 4 * (lane >> 4), c = full + (lane & 15);
            if (r < rows && c < cn) {
                const int rb = vf[r], cnt = vn[r];
                const double* vwr = vw + r * MAXT;
                double acc = 0.0;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {               // taps beyond the row's count: weight 0.0 (stored so) times a 0.0 sample
                    const bool on = t < cnt;
                    const float xv = raw[(rb + (on ? t : 0)) * LW + c];
                    acc = acc + vwr[t] * (double)(on ? xv : 0.f);
                }
                mid[r * LW + c] = (double)(float)acc;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        auto horizontal = [&](auto ntaps) {
            constexpr int NT = decltype(ntaps)::value;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int r = wv + 4 * k;
                if (r >= rows) break;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double xs[NT];         // intermediate samples are stored widened: one conversion per sample, not per tap
#pragma unroll
                    for (int t = 0; t < NT; ++t) xs[t] = hp[j][t][4 * k * LW];
                    double acc = 0.0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc = acc + w[j][t] * xs[t];
                    o[j] = (float)acc;
                }
                float* dr = a.dst + (size_t)(ry0 + r) * a.dst_w;
                if (a.vec) {
                    *reinterpret_cast<float2*>(dr + xa) = make_float2(o[0], o[1]);
                    if (xb < a.dst_w) *reinterpret_cast<float2*>(dr + xb) = make_float2(o[2], o[3]);
                } else {
                    if (xa < a.dst_w) dr[xa] = o[0];
                    if (xa + 1 < a.dst_w) dr[xa + 1] = o[1];
                    if (xb < a.dst_w) dr[xb] = o[2];
                    if (xb + 1 < a.dst_w) dr[xb + 1] = o[3];
                }
            }
        };
        if (xa < a.dst_w) {
            if (short_taps) horizontal(std::integral_constant<int, (MAXT > 1 ? MAXT - 1 : 1)>{});
            else horizontal(std::integral_constant<int, MAXT>{});
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// =============================================================================================
// conv12 with exact products from the matrix instruction (the production layer-1+2 kernel).
//
// Strict mode may not fuse the multiply into the add, so on the VALU every MAC costs two
// instructions.  gfx950's K=1 multi-block MFMA computes D = A (x) B + C per block with one
// rounding (an fmaf); with C = 0 that is exactly the correctly rounded fp32 PRODUCT.  So the
// matrix pipe produces the 64-channel x 32-pixel outer product of one tap
//     D[k][px] = round(w1[k][tap] * Y[px + tap])          (v_mfma_f32_32x32x1_2b_f32, C = 0)
// and the VALU only accumulates  acc[k][px] += D[k][px]  (16 v_pk_add_f32), in the reference's
// tap order.  (Measured: the fp32 MFMA does not overlap with VALU work on gfx950 -- it is simply the cheapest way
// to get 2048 correctly rounded products: 64 cycles, two one-register operands, no SGPR weight traffic.)  Elementwise adds do not care about the MFMA register layout; it matters only for the
// bias/ReLU epilogue and the hand-off to layer 2:
//     block b = reg/16 (lanes 32b..32b+31 feed A/B of block b), row = 8*((reg%16)/4) + 4*(lane/32) + reg%4,
//     col = lane%32.                        A_b[row] <- lane 32b+row,   B_b[col] <- lane 32b+col
// Layer 1: block b carries channels 32b..32b+31, both blocks see the same 32 pixels.
// Layer 2: ReLU(c1) goes through a per-wave LDS slab [f][px]; one MFMA then multiplies channel
//     pair (f, f+1) (block 0 / block 1) against all 32 outputs m, and the VALU adds block 0 then
//     block 1 into acc2 -- channel order 0..63 preserved.
// FAST mode chains C = acc instead (an FMA chain on the matrix pipe, no VALU adds).
// =============================================================================================
// PIN(v): an empty volatile asm that consumes and redefines v.  Instruction selection otherwise treats
// the pure adds / MFMAs as freely movable and sinks or hoists them across the software pipeline.
#define PIN(v) asm volatile("" : "+v"(v))
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int M_NW = 8;                            // waves per workgroup (512 threads; 2 workgroups per CU = 4 waves per SIMD)
constexpr int M_TW = 64, M_TH = 2 * M_NW;          // block tile 64 x 16: 2 segments wide, 16 rows -> 4 segments per wave
constexpr int M_LW = M_TW + 8;
constexpr int M_W1 = 81 * 64, M_W2 = 32 * 64, M_B1 = 64, M_B2 = 32;
constexpr int M_C1 = 32 * 32;                      // per-wave layer-1 slab: 32 channels x 32 px (one MFMA block at a time)
constexpr int M_YT = (M_TH + 8) * M_LW;            // one Y tile with its 4-sample halo
constexpr int M_BPC = 2;                           // resident workgroups per CU
static_assert(M_TH == kConv12TileRows && M_BPC == kConv12BlocksPerCU, "srcnn_kernels.h carries this geometry for the band planners");
constexpr int m_ybufs(bool ld) { return ld ? 2 : 1; }          // LDS-DMA staging double-buffers the Y tile
constexpr int m_lds_floats(bool ld) { return M_W1 + M_W2 + M_B1 + M_B2 + m_ybufs(ld) * M_YT + M_NW * M_C1 + 2; }   // + the two tile-queue slots

// One tap-step = one MFMA (products) whose result the wave's own adds then wait for; the other three waves of the SIMD fill
// the gap (one result buffer: 126 VGPRs -> 4 waves per SIMD.  The software-pipelined two-buffer forms and the 256-thread
// geometries of rounds 1-4 measured 2-5 % slower and are gone: docs/HISTORY.md 4.1).
// LD: weights and Y tiles staged by LDS-DMA one tile ahead (production); LD = false is the same geometry with the
// load -> wait -> ds_write staging it replaced (SRCNN_CONV12_DMA=0: the fallback should the DMA path ever be suspected).
// RELAX: bit 0 = layer 1, bit 1 = layer 2 evaluated as FMA chains on the matrix pipe (C = acc: one rounding per tap instead of
// the reference's two).  0 = STRICT (production, bit-exact), 3 = SRCNN_MODE_FAST; 1 and 2 exist for the per-layer error
// matrix (profiles/r04_error_matrix.txt) and the SRCNN_MODE_RELAXED experiments.
template <int RELAX, bool LD>
__global__ __launch_bounds__(64 * M_NW, 4) void k_conv12_mfma(
    const float* __restrict__ Y, int W, int H, int y_row_base, int y_rows,
    float* __restrict__ C2, size_t plane_stride, int out_row0, int out_rows, int tiles_x, int ntiles,
    unsigned long long* __restrict__ clk, unsigned* __restrict__ queue)
{
    constexpr int NW = M_NW, NT = 64 * NW, TH = M_TH, YT = M_YT;
    // clk != NULL (srcnn_debug_clock_probe): workgroup 0 -- resident from the first round to the last -- stamps the shader
    // clock counter and the constant 100 MHz counter when it starts and when it ends: their ratio is the clock this launch ran at
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk && blockIdx.x == 0) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr bool STRICT1 = !(RELAX & 1), STRICT2 = !(RELAX & 2);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* W1s = lds;                    // [tap][lane]: lane = channel
    float* W2s = W1s + M_W1;             // [f/2][lane]: lanes 0-31 -> w2[m=lane][f], 32-63 -> w2[m=lane-32][f+1]
    float* B1s = W2s + M_W2;             // [half][reg] layer-1 bias in accumulator layout
    float* B2s = B1s + M_B1;             // [half][reg] layer-2 bias in accumulator layout
    float* Yt  = B2s + M_B2;             // [m_ybufs][YT]
    float* C1s = Yt + m_ybufs(LD) * YT;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5, col = lane & 31;

    // The block's weight image (29 KB, MFMA operand order).  With the production geometry it is moved by LDS-DMA as well: all
    // 15 pieces per thread are in flight at once and land before the first tile's barrier; the load -> wait -> ds_write loops
    // cost ten global-memory latencies per launch, which a band of a ProcessSRCNN call or a small image cannot amortise.
    constexpr bool WDMA = LD;
    const int wave_e0w = __builtin_amdgcn_readfirstlane(tid & ~63);
    if constexpr (WDMA) {
#pragma unroll
        for (int i = 0; i < (M_W1 + NT - 1) / NT; ++i) {
            const int e = tid + i * NT;
            if (e < M_W1) rs_dma_dword(&cW.w1t[0][0] + e, W1s + i * NT + wave_e0w);
        }
#pragma unroll
        for (int i = 0; i < (M_W2 + NT - 1) / NT; ++i) {
            const int e = tid + i * NT;
            if (e < M_W2) { const int fp = e >> 6, l = e & 63; rs_dma_dword(&cW.w2[l & 31][2 * fp + (l >> 5)], W2s + i * NT + wave_e0w); }
        }
        if (tid < 64) {
            const int hf = tid >> 5, r = tid & 31;
            rs_dma_dword(&cW.b1[32 * (r >> 4) + 8 * ((r & 15) >> 2) + 4 * hf + (r & 3)], B1s);
        }
        if (tid < 32) {
            const int hf = tid >> 4, r = tid & 15;
            rs_dma_dword(&cW.b2[8 * (r >> 2) + 4 * hf + (r & 3)], B2s);
        }
    } else {
        for (int e = tid; e < M_W1; e += NT) W1s[e] = (&cW.w1t[0][0])[e];
        for (int e = tid; e < M_W2; e += NT) {
            const int fp = e >> 6, l = e & 63;
            W2s[e] = cW.w2[l & 31][2 * fp + (l >> 5)];
        }
        if (tid < 64) {
            const int hf = tid >> 5, r = tid & 31;
            B1s[tid] = cW.b1[32 * (r >> 4) + 8 * ((r & 15) >> 2) + 4 * hf + (r & 3)];
        }
        if (tid < 32) {
            const int hf = tid >> 4, r = tid & 15;
            B2s[tid] = cW.b2[8 * (r >> 2) + 4 * hf + (r & 3)];
        }
    }
    float* myC1 = C1s + wv * M_C1;
    const f32x32 zero32 = {};

    // Persistent grid, static stride.  The tiles that fill whole rounds of the grid are taken whole (four segments per
    // wave); the tiles of the last, partly filled round are dealt out in QUARTERS (one segment per wave), so that round costs
    // ceil(4 * left / grid) quarter-steps instead of one whole step: an 8K frame's 144 leftover tiles take half a tile-time
    // on 512 workgroups instead of leaving 368 of them idle for a whole one (-0.8 % of the launch).
    const int full = (ntiles / (int)gridDim.x) * (int)gridDim.x;
    const int nitems = full + 4 * (ntiles - full);
    // The Y tile (+4 halo) of the NEXT item is on its way while the current one is computed: LDS-DMA (global_load_lds_dword, no
    // VGPR destination -- the kernel has none to spare) into the other of two tile buffers; element e of a tile goes to
    // Yt[e], and consecutive lanes take consecutive e, which is exactly the DMA's "wave base + lane * 4" destination.  The
    // staging used to be a load -> wait -> ds_write loop between two barriers: NPRE global-memory latencies per tile with all
    // of the workgroup's waves standing still, and two barriers per tile instead of one.
    constexpr bool DMA = LD;
    constexpr int NPRE = (YT + NT - 1) / NT;
    auto tile_of = [&](int item, int& s0, int& s1) {
        int tile = item; s0 = 0; s1 = 4;
        if (item >= full) { const int i = item - full; tile = full + (i >> 2); s0 = i & 3; s1 = s0 + 1; }
        return tile;
    };
    const int wave_e0 = wave_e0w;         // first element of this wave's 64 (an SGPR: the DMA's LDS base must be uniform)
    auto request = [&](int item, float* buf) {
        int q0, q1;
        const int tile = tile_of(item, q0, q1);
        const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
        const int tx0 = txi * M_TW, ty0 = out_row0 + tyi * TH;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int e = tid + i * NT;
            if (e < YT) {
                const int r = e / M_LW, c = e - r * M_LW;
                const int gy = clampi(clampi(ty0 + r - 4, 0, H - 1), y_row_base, y_row_base + y_rows - 1), gx = clampi(tx0 + c - 4, 0, W - 1);
                rs_dma_dword(Y + (size_t)(gy - y_row_base) * W + gx, buf + i * NT + wave_e0);
            }
        }
    };
    // Which item next?  Static stride (queue == NULL): item + gridDim.x.  Tile QUEUE (the production form with LDS-DMA staging):
    // a workgroup's first item is its own index, every further one comes from a global counter (index = gridDim.x + old value).
    // Why: the SIMDs issue oldest-wave-first, so of the two workgroups that share a CU the one that arrived first runs ahead
    // -- measured, it finishes its static half of the tiles after 4.1 ms of a 7.3 ms launch (profiles/r04_process_clock.txt) --
    // and the rest of the launch runs at two waves per SIMD instead of four.  With the queue both keep drawing tiles until
    // none are left.  The index is needed one item AHEAD (the next tile's DMA is issued at the top of the current item):
    // lane 0 draws it an item early and passes it through a two-slot LDS mailbox that the item's one barrier publishes.
    // The counters clean up after themselves: the last workgroup to leave puts both back to zero (queue[1] counts leavers),
    // so a workspace's queue needs no memset between the launches of its stream.
    int* qslot = reinterpret_cast<int*>(C1s + NW * M_C1);
    const bool dyn = DMA && queue != nullptr;
    int it = 0;
    int item = blockIdx.x;
    if constexpr (DMA) {
        if (item < nitems) request(item, Yt);
        if (dyn && tid == 0 && item < nitems) qslot[0] = (int)gridDim.x + (int)atomicAdd(queue, 1u);
    }
    for (; item < nitems; ++it) {
        int s0, s1;
        const int tile = tile_of(item, s0, s1);
        const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
        const int tx0 = txi * M_TW, ty0 = out_row0 + tyi * TH;
        const float* Ytc = Yt;
        int next_item = item + (int)gridDim.x;
        int drawn = 0x7fffffff;
        if constexpr (DMA) {
            Ytc = Yt + (it & 1) * YT;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the item's tile has landed
            __syncthreads();                                    // ... everybody's has; the previous item is finished everywhere
            if (dyn) next_item = __builtin_amdgcn_readfirstlane(qslot[it & 1]);
            if (next_item < nitems) request(next_item, Yt + ((it + 1) & 1) * YT);
            if (dyn && tid == 0 && next_item < nitems) drawn = (int)gridDim.x + (int)atomicAdd(queue, 1u);     // the item after next
        } else {
            __syncthreads();
            for (int e = tid; e < YT; e += NT) {
                const int r = e / M_LW, c = e - r * M_LW;
                const int gy = clampi(clampi(ty0 + r - 4, 0, H - 1), y_row_base, y_row_base + y_rows - 1), gx = clampi(tx0 + c - 4, 0, W - 1);
                Yt[e] = Y[(size_t)(gy - y_row_base) * W + gx];
            }
            __syncthreads();
        }

#pragma unroll 1
        for (int s = s0; s < s1; ++s) {
            const int sg = wv * 4 + s;
            const int trow = sg >> 1, seg = sg & 1;
            const float* yrow = Ytc + trow * M_LW + seg * 32 + col;

            // ---- layer 1: 81 taps, one MFMA (products) + 16 packed adds each ----
            // Pinned with sched_barrier + PIN: left alone, the scheduler hoists all 81 independent MFMAs and
            // spills their 32-register results.  Operands are fetched from LDS two taps ahead.
            f32x32 acc = zero32;
            if constexpr (!STRICT1) {
#pragma unroll
                for (int t = 0; t < 81; ++t) {
                    const float a = W1s[t * 64 + lane];
                    const float b = yrow[(t / 9) * M_LW + (t % 9)];
                    acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, acc, 0, 0, 0);
                }
            } else {
                float a1 = W1s[lane], b1 = yrow[0];
                float a2 = W1s[64 + lane], b2 = yrow[1];
#pragma unroll
                for (int t = 0; t < 81; ++t) {
                    f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a1, b1, zero32, 0, 0, 0);
                    PIN(d);
                    a1 = a2; b1 = b2;
                    if (t + 2 < 81) {
                        a2 = W1s[(t + 2) * 64 + lane];
                        b2 = yrow[((t + 2) / 9) * M_LW + ((t + 2) % 9)];
                    }
                    // tap 0: the reference's "0 + p" is p itself (a -0 product differs from the reference's +0 only
                    // until the first non-(-0) term or the bias add, never in the stored value), so the first
                    // result simply becomes the accumulator
                    if (t == 0) acc = d; else acc += d;
                    PIN(acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- bias + ReLU, then layer 2.  The 64 channels go through the wave's LDS slab [f][px] one MFMA
            //      block (32 channels) at a time, so the slab is only 4 KB per wave. ----
            f32x16 acc2 = {};
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = fmaxf(acc[16 * blk + r] + B1s[half * 32 + 16 * blk + r], 0.f);
                    const int f = 8 * (r >> 2) + (r & 3);                  // + 4*half : channel within the block
                    myC1[(f + 4 * half) * 32 + col] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

                // 16 MFMAs per block, each = channels (f, f+1) x 32 outputs x 32 pixels; A = weights (rows = m),
                // B = activations (cols = px); block 0 of the result is channel f, block 1 channel f+1.
                const float* w2p = W2s + blk * 16 * 64;
                float a1 = w2p[lane], b1 = myC1[lane];
                float a2 = w2p[64 + lane], b2 = myC1[64 + lane];
                if constexpr (!STRICT2) {
                    // FAST: the slab/weight layout (lanes 0-31 channel f, lanes 32-63 channel f+1) is exactly the
                    // K=2 operand layout of the single-block MFMA, so the pair is one chained FMA update of acc2.
#pragma unroll
                    for (int fp = 0; fp < 16; ++fp)
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2p[fp * 64 + lane], myC1[fp * 64 + lane], acc2, 0, 0, 0);
                } else {
#pragma unroll
                    for (int fp = 0; fp < 16; ++fp) {
                        f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a1, b1, zero32, 0, 0, 0);
                        PIN(d);
                        a1 = a2; b1 = b2;
                        if (fp + 2 < 16) { a2 = w2p[(fp + 2) * 64 + lane]; b2 = myC1[(fp + 2) * 64 + lane]; }
                        if (blk == 0 && fp == 0) acc2 = __builtin_shufflevector(d, d, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                        else
                        acc2 += __builtin_shufflevector(d, d, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                        acc2 += __builtin_shufflevector(d, d, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31);
                        PIN(acc2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_wave_barrier();     // slab is rewritten by the next block / segment
            }

            // ---- bias + ReLU + store: reg r -> output m = 8*(r/4) + 4*half + r%4, pixel col ----
            const int row = ty0 + trow, x = tx0 + seg * 32 + col;
            if (row < out_row0 + out_rows && row < H && x < W) {
                float* dst = C2 + (size_t)(row - out_row0) * W + x;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = 8 * (r >> 2) + (r & 3);      // + 4*half
                    dst[(size_t)(m + 4 * half) * plane_stride] = fmaxf(acc2[r] + B2s[half * 16 + r], 0.f);
                }
            }
        }
        if (dyn && tid == 0) qslot[(it + 1) & 1] = drawn;        // read by everybody after the next item's barrier
        item = next_item;
    }
    if (dyn && tid == 0) {
        // every draw of this workgroup has returned (its value went through the mailbox); the last one out resets the queue
        if (atomicAdd(queue + 1, 1u) == gridDim.x - 1) { queue[0] = 0u; queue[1] = 0u; }
    }
    if (clk && blockIdx.x == 0) {
        __syncthreads();
        if (threadIdx.x == 0) {
            clk[0] = __builtin_amdgcn_s_memtime() - clk_c0;
            clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
        }
    }
}

// =============================================================================================
// conv3: 5x5x32->1 + bias + clamp[0,255].  Per channel the reference sums fp32 products in an
// fp64 accumulator (25 taps, row-major over the window), then folds it into an fp32 running sum:
//     v_pk_mul_f32 (2 products) ; v_cvt_f64_f32 ; v_add_f64      -- 2.5 VALU instructions per MAC.
// Lane = 4 vertically adjacent pixels of one column (8x5 window per channel in registers).
// The c2 tile (+2 halo, clamp-to-edge of the ACTIVATIONS at the true image border) goes through a
// double-buffered LDS stage, MC channels at a time: the global loads of chunk c+1 are issued before
// the arithmetic of chunk c and land in LDS after it.  All staging index math is shifts/compares
// (no div/mod): 64 body columns by (lane, wave) and the 4 halo columns by a second small pass.
// =============================================================================================
constexpr int C3_TW = 64, C3_TH = 16;
constexpr int C3_LW = C3_TW + 4, C3_LH = C3_TH + 4;
constexpr int C3_CH = C3_LH * C3_LW;               // floats per staged channel
static_assert(C3_LH % 4 == 0, "row slots");
// channels per double-buffered LDS stage: 2 with DMA staging (25.6 KB of LDS and 76 VGPRs: six workgroups per CU), 4 with
// register staging (47.4 KB: three)
constexpr int c3_mc(bool sdma) { return sdma ? 2 : 4; }

// X64 (experiment, SRCNN_MODE_RELAXED bit 2): the products are formed exactly -- v_fma_f64 on widened operands -- instead of
// being rounded to fp32 first; everything else (per-channel fp64 sum in window order, fp32 running sum over the channels,
// bias, clamp) is the reference's.  Per channel and lane 40 v_cvt_f64_f32 + 100 v_fma_f64 instead of 50 v_pk_mul_f32 +
// 104 v_cvt_f64_f32 + 100 v_add_f64.
// SDMA (round 6, the production form of the strict kernel): the 64 body columns of every staged row go global -> LDS by
// LDS-DMA ("uniform plane base + per-lane 32-bit offset" addressing: no VGPR destination, no ds_write, no address arithmetic
// on the VALU); only the 4 halo columns still pass through registers; the lane's window is read from two fixed LDS row bases.
// What that buys is REGISTERS, not instructions: at the old 4-channel stage the kernel ran no faster with 30 fewer index /
// staging instructions per chunk (2.33 vs 2.30 ms, profiles/r06_conv3_ab.txt -- they were hidden, as the v_movs of round 2
// had been), but at 76 VGPRs and 2-channel stages (25.6 KB of LDS) six workgroups fit a CU instead of three, and v_cvt_f64_f32
// / v_add_f64 issue closer to their rate with six waves per SIMD: 2.24 vs 2.30 ms per 8K frame, the whole path -1 %.
// 1-channel stages (32 barriers per tile) and 7-8 waves per SIMD (spills) are slower: 3.1 / 5.4 / 8.3 ms.
// SDMA = false (SRCNN_CONV3_WDMA=0): the register-staged 4-channel form it replaced.
template <bool STRICT, bool OFF64 = false, bool X64 = false, bool SDMA = false>
__global__ __launch_bounds__(256, SDMA ? 6 : 3) void k_conv3(
    const float* __restrict__ C2, size_t plane_stride, int W, int H, int c2_row_base, int c2_rows,
    float* __restrict__ out, int out_row0, int out_rows, int weights_by_dma)
{
    constexpr int C3_MC = c3_mc(SDMA);
    constexpr int C3_BODY = C3_MC * C3_LH / 4;         // body loads per thread per chunk (4 row slots)
    constexpr int C3_HALO = (C3_MC * C3_LH * 4 + 255) / 256;
    __shared__ float tile[2][C3_MC * C3_CH];
    // strict: [m][dy][6]: (w0,w1) (w2,w3) w4 pad -- packed-operand order;  X64: [m][dy][dx] as doubles
    __shared__ __attribute__((aligned(16))) float w3s[X64 ? C2N * 50 : C2N * 30];

    const int tx0 = blockIdx.x * C3_TW;
    const int ty0 = out_row0 + blockIdx.y * C3_TH;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c2_last = c2_row_base + c2_rows - 1;
    if constexpr (X64) {
        double* w3d = reinterpret_cast<double*>(w3s);
        for (int e = tid; e < C2N * 25; e += 256) w3d[e] = (double)(&cW.w3[0][0])[e];
    } else if constexpr (STRICT) {
        // the packed weight image exists in constant memory as such (filled on the host): four LDS-DMA pieces per thread, all in
        // flight at once -- every one of the 32 400 workgroups of an 8K frame pays this prologue
        const int wave_e0 = __builtin_amdgcn_readfirstlane(tid & ~63);
        if (SDMA || weights_by_dma) {
#pragma unroll
            for (int i = 0; i < (C2N * 30 + 255) / 256; ++i) {
                const int e = tid + i * 256;
                if (e < C2N * 30) rs_dma_dword(cW.w3p + e, w3s + i * 256 + wave_e0);
            }
        } else {                                               // SRCNN_CONV3_WDMA=0 (A/B runs, fallback)
            for (int e = tid; e < C2N * 30; e += 256) w3s[e] = cW.w3p[e];
        }
    }

    // staging geometry (fixed per thread)
    const int bx = min(tx0 + lane, W - 1);                                  // body column
    const int hrow = tid >> 2, hq = tid & 3;                                // halo: 4 columns per staged row
    const int hx = clampi(tx0 + (hq < 2 ? hq - 2 : 62 + hq), 0, W - 1);     // image cols tx0-2,tx0-1,tx0+64,tx0+65
    const int hlc = hq < 2 ? hq : 64 + hq;                                  // LDS cols 0,1,66,67

    float body[C3_BODY], halo[C3_HALO];

    auto row_of = [&](int r) {                // staged row r (0..LH-1) -> row inside the c2 buffer
        int gy = clampi(ty0 + r - 2, 0, H - 1);
        gy = clampi(gy, c2_row_base, c2_last);
        return gy - c2_row_base;
    };
    // Where this thread reads inside a plane never changes during the tile: 32-bit element offsets, computed once.  The loads
    // are then "uniform plane base (SGPR pair) + per-lane 32-bit offset" -- no 64-bit address arithmetic on the VALU per load
    // (it was 42 v_lshl_add_u64 per 4-channel chunk, 4 % of the kernel's vector instructions).  OFF64: the launcher's choice for
    // planes of 4 GiB or more (stage-level calls on caller-provided planes; the Y path bands long before that).
    using offs_t = std::conditional_t<OFF64, size_t, unsigned>;
    offs_t boff[C3_LH / 4], hoff[C3_HALO];
    int hm[C3_HALO];
#pragma unroll
    for (int j = 0; j < C3_LH / 4; ++j) boff[j] = ((offs_t)row_of(4 * j + wv) * (offs_t)W + (offs_t)bx) * 4u;     // BYTE offsets
#pragma unroll
    for (int i = 0; i < C3_HALO; ++i) {
        const int rr = min(hrow + 64 * i, C3_MC * C3_LH - 1);
        hm[i] = rr / C3_LH;
        hoff[i] = ((offs_t)row_of(rr - hm[i] * C3_LH) * (offs_t)W + (offs_t)hx) * 4u;
    }
    const int wv_s = __builtin_amdgcn_readfirstlane(wv);
    // SDMA halo: thread (row, hq) of the first 80 moves its 4 halo columns of one staged row, once per channel of the chunk
    // (uniform plane base again); waves 2 and 3 skip it
    float halo4[C3_MC];
    const int h4row = min(tid >> 2, C3_LH - 1);
    const unsigned h4off = ((unsigned)row_of(h4row) * (unsigned)W + (unsigned)hx) * 4u;
    auto issue_dma = [&](int mc, float* dst) {     // SDMA: global -> LDS for the chunk starting at channel mc
        if (tid < C3_LH * 4) {
#pragma unroll
            for (int m = 0; m < C3_MC; ++m)
                halo4[m] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(C2 + (size_t)(mc + m) * plane_stride) + h4off);
        }
#pragma unroll
        for (int m = 0; m < C3_MC; ++m) {
            // the plane's base as an SGPR pair, whatever the halo loads above made of the same expression
            const unsigned long long pb = reinterpret_cast<unsigned long long>(C2 + (size_t)(mc + m) * plane_stride);
            const unsigned ub_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pb >> 32));
            const unsigned ub_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)pb);       // (the builtin returns int: no sign extension)
            const unsigned long long ub = ((unsigned long long)ub_hi << 32) | (unsigned long long)ub_lo;
#pragma unroll
            for (int j = 0; j < C3_LH / 4; ++j) {
                const int i = m * (C3_LH / 4) + j;
                if constexpr (!OFF64)
                    rs_dma_dword_s(reinterpret_cast<const void*>(ub), boff[j], dst + (wv_s + 4 * i) * C3_LW + 2);
            }
        }
    };
    auto land_dma = [&](float* dst) {              // SDMA: the halo registers -> LDS; every DMA of this wave has landed
        if (tid < C3_LH * 4) {
#pragma unroll
            for (int m = 0; m < C3_MC; ++m) dst[m * C3_CH + h4row * C3_LW + hlc] = halo4[m];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto issue = [&](int mc) {                // global -> registers for chunk starting at channel mc
#pragma unroll
        for (int i = 0; i < C3_BODY; ++i) {
            // staged row rr = 4*i + wv; LH is a multiple of 4, so the channel is i / (LH/4) at compile time
            const int m = i / (C3_LH / 4), j = i % (C3_LH / 4);
            const char* plane = reinterpret_cast<const char*>(C2 + (size_t)(mc + m) * plane_stride);      // wave-uniform
            body[i] = *reinterpret_cast<const float*>(plane + boff[j]);
        }
#pragma unroll
        for (int i = 0; i < C3_HALO; ++i) {
            const int rr = hrow + 64 * i;
            if (rr < C3_MC * C3_LH)
                halo[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(C2 + ((size_t)mc + hm[i]) * plane_stride) + hoff[i]);
        }
    };
    auto land = [&](float* dst) {             // registers -> LDS
#pragma unroll
        for (int i = 0; i < C3_BODY; ++i) dst[(wv + 4 * i) * C3_LW + 2 + lane] = body[i];
#pragma unroll
        for (int i = 0; i < C3_HALO; ++i) {
            const int rr = hrow + 64 * i;
            if (rr < C3_MC * C3_LH) dst[rr * C3_LW + hlc] = halo[i];
        }
    };

    float sum[4] = {0.f, 0.f, 0.f, 0.f};

    if constexpr (SDMA) { issue_dma(0, tile[0]); land_dma(tile[0]); }
    else { issue(0); land(tile[0]); }
    if constexpr (STRICT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the weight image's DMA
    __syncthreads();

    // SDMA: the lane's two LDS row bases (rows 0-3 and 4-7 of its 8-row window; every ds_read offset then fits the
    // instruction's 8-bit dword field).  The second one is made opaque so that it stays a base instead of being folded into
    // ten per-channel address additions.
    const int offA = (wv * 4) * C3_LW + lane;
    int offB = offA + 4 * C3_LW;
    asm volatile("" : "+v"(offB));

#pragma unroll 1
    for (int c = 0; c < C2N / C3_MC; ++c) {
        const float* cur = tile[c & 1];
        if (c + 1 < C2N / C3_MC) {
            if constexpr (SDMA) issue_dma((c + 1) * C3_MC, tile[(c + 1) & 1]);
            else issue((c + 1) * C3_MC);
        }
#pragma unroll 1
        for (int m = 0; m < C3_MC; ++m) {
            const float* t = cur + m * C3_CH + (wv * 4) * C3_LW + lane;
            if constexpr (X64) {
                const double* wd = reinterpret_cast<const double*>(w3s) + (c * C3_MC + m) * 25;
                double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    double v[5];
#pragma unroll
                    for (int cc = 0; cc < 5; ++cc) v[cc] = (double)t[r * C3_LW + cc];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int dy = r - q;                       // window row of pixel q that staged row r is
                        if (dy >= 0 && dy < 5) {
#pragma unroll
                            for (int dx = 0; dx < 5; ++dx) a[q] = __builtin_fma(wd[dy * 5 + dx], v[dx], a[q]);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) sum[q] = (float)((double)sum[q] + a[q]);
            } else if constexpr (STRICT && SDMA) {
                // As the branch below, with the window rows addressed from two fixed bases and the fifth window column
                // read as the (row k, row k+1) pairs its packed products consume -- pixels (q, q+1) share a weight there.
                typedef float f2 __attribute__((ext_vector_type(2)));
                const float* tA = cur + m * C3_CH + offA;
                const float* tB = cur + m * C3_CH + offB;
                f2 va[8], vb[8], vp[7];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    va[r] = f2{tA[r * C3_LW + 0], tA[r * C3_LW + 1]};
                    vb[r] = f2{tA[r * C3_LW + 2], tA[r * C3_LW + 3]};
                    va[r + 4] = f2{tB[r * C3_LW + 0], tB[r * C3_LW + 1]};
                    vb[r + 4] = f2{tB[r * C3_LW + 2], tB[r * C3_LW + 3]};
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    vp[k] = f2{tA[k * C3_LW + 4], tA[(k + 1) * C3_LW + 4]};
                    vp[k + 4] = f2{tB[k * C3_LW + 4], tB[(k + 1) * C3_LW + 4]};
                }
                vp[3] = f2{tA[3 * C3_LW + 4], tB[4]};
                const float* wrow = w3s + (c * C3_MC + m) * 30;
                double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int dy = 0; dy < 5; ++dy) {
                    const f2 wa = f2{wrow[dy * 6 + 0], wrow[dy * 6 + 1]};
                    const f2 wb = f2{wrow[dy * 6 + 2], wrow[dy * 6 + 3]};
                    const f2 wc = f2{wrow[dy * 6 + 4], wrow[dy * 6 + 4]};
                    const f2 pc01 = wc * vp[dy], pc23 = wc * vp[dy + 2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f2 pa = wa * va[q + dy], pb = wb * vb[q + dy];
                        const float pc = q == 0 ? pc01.x : q == 1 ? pc01.y : q == 2 ? pc23.x : pc23.y;
                        a[q] = (dy == 0) ? (double)pa.x : a[q] + (double)pa.x;
                        a[q] = a[q] + (double)pa.y;
                        a[q] = a[q] + (double)pb.x;
                        a[q] = a[q] + (double)pb.y;
                        a[q] = a[q] + (double)pc;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) sum[q] = (float)((double)sum[q] + a[q]);
            } else if constexpr (STRICT) {
                // Products two at a time (v_pk_mul_f32) with BOTH operands already sitting in register pairs: the window
                // row as (c0,c1) (c2,c3) c4 straight from ds_read2_b32, the channel's weights in the same shape from the
                // block's LDS copy (uniform address -> broadcast read).  Round 2 took the weights from SGPRs, which cost
                // a v_mov_b32 per packed operand (53 of 317 VALU instructions per channel).  Measured: the same 2.31 ms per
                // 8K frame with or without them -- the pace is set by the 104 v_cvt_f64_f32 + 100 v_add_f64 + 50
                // v_pk_mul_f32 per channel at their measured issue rates (4.4 / 5.5 / 4.4 cycles, profiles/r01_valu_rates.txt:
                // 1305 cycles per wave-channel predicted, 1298 measured); the movs rode in their shadow.  2-channel stages
                // with 4 waves per SIMD: 2.48 ms (more barriers, spills).
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 va[8], vb[8];
                float vc[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    va[r] = f2{t[r * C3_LW + 0], t[r * C3_LW + 1]};
                    vb[r] = f2{t[r * C3_LW + 2], t[r * C3_LW + 3]};
                    vc[r] = t[r * C3_LW + 4];
                }
                const float* wrow = w3s + (c * C3_MC + m) * 30;
                // four independent fp64 chains (one per pixel) advance tap by tap, so consecutive v_add_f64 never
                // depend on each other
                double a[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int dy = 0; dy < 5; ++dy) {
                    const f2 wa = f2{wrow[dy * 6 + 0], wrow[dy * 6 + 1]};
                    const f2 wb = f2{wrow[dy * 6 + 2], wrow[dy * 6 + 3]};
                    const float wc = wrow[dy * 6 + 4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f2 pa = wa * va[q + dy], pb = wb * vb[q + dy];
                        const float pc = wc * vc[q + dy];
                        // tap (0,0): "0.0 + p" is p (a -0 differs from the reference's +0 only until it is folded
                        // into the fp32 running sum, which can never be -0)
                        a[q] = (dy == 0) ? (double)pa.x : a[q] + (double)pa.x;
                        a[q] = a[q] + (double)pa.y;
                        a[q] = a[q] + (double)pb.x;
                        a[q] = a[q] + (double)pb.y;
                        a[q] = a[q] + (double)pc;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) sum[q] = (float)((double)sum[q] + a[q]);
            } else {
                const float* wm = cW.w3[c * C3_MC + m];
                float win[8][5];
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int cc = 0; cc < 5; ++cc) win[r][cc] = t[r * C3_LW + cc];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float a = 0.f;
#pragma unroll
                    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 5; ++dx) a = __builtin_fmaf(wm[dy * 5 + dx], win[q + dy][dx], a);
                    sum[q] += a;
                }
            }
        }
        if (c + 1 < C2N / C3_MC) {
            if constexpr (SDMA) land_dma(tile[(c + 1) & 1]);
            else land(tile[(c + 1) & 1]);
        }
        __syncthreads();
    }
    const int x = tx0 + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = ty0 + wv * 4 + q;
        if (x < W && row < H && row < out_row0 + out_rows) {
            float v = sum[q] + cW.b3;
            v = fminf(fmaxf(v, 0.f), 255.f);
            out[(size_t)(row - out_row0) * W + x] = v;
        }
    }
}

#ifndef SRCNN_STRICT_ONLY
// =============================================================================================
// conv3 for the non-parity tiers: plain fp32 FMA chains.  At 2 VALU cycles per 64 FMAs the strict kernel's
// one-column-per-lane window (40 ds_read_b32 per 100 FMAs) is LDS-bound, so here a lane owns 2 adjacent
// columns x 4 rows: its 8x6 window is 24 aligned ds_read_b64 per 200 FMAs.  Tile 128 x 16, 2 channels per
// double-buffered LDS stage.
// =============================================================================================
constexpr int CF_TW = 128, CF_TH = 16, CF_MC = 2;
constexpr int CF_LW = CF_TW + 4, CF_LH = CF_TH + 4;
constexpr int CF_CH = CF_LH * CF_LW;
constexpr int CF_BODY = CF_MC * CF_LH / 2;          // body loads per thread per chunk (2 row slots of 128 columns)

__global__ __launch_bounds__(256) void k_conv3_fast(
    const float* __restrict__ C2, size_t plane_stride, int W, int H, int c2_row_base, int c2_rows,
    float* __restrict__ out, int out_row0, int out_rows)
{
    __shared__ __attribute__((aligned(16))) float tile[2][CF_MC * CF_CH];

    const int tx0 = blockIdx.x * CF_TW;
    const int ty0 = out_row0 + blockIdx.y * CF_TH;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c2_last = c2_row_base + c2_rows - 1;

    const int bcol = tid & 127, bslot = tid >> 7;                       // body: 128 columns x 2 row slots
    const int bx = min(tx0 + bcol, W - 1);
    const int hrow = tid >> 2, hq = tid & 3;                            // halo: 4 columns per staged row
    const int hx = clampi(tx0 + (hq < 2 ? hq - 2 : 126 + hq), 0, W - 1);
    const int hlc = hq < 2 ? hq : 128 + hq;

    float body[CF_BODY], halo = 0.f;
    auto row_of = [&](int r) {
        int gy = clampi(ty0 + r - 2, 0, H - 1);
        gy = clampi(gy, c2_row_base, c2_last);
        return gy - c2_row_base;
    };
    auto issue = [&](int mc) {
#pragma unroll
        for (int i = 0; i < CF_BODY; ++i) {
            const int m = i / (CF_LH / 2), r = 2 * (i % (CF_LH / 2)) + bslot;      // staged row 2*i + bslot
            body[i] = C2[(size_t)(mc + m) * plane_stride + (size_t)row_of(r) * W + bx];
        }
        if (hrow < CF_MC * CF_LH) {
            const int m = hrow / CF_LH, r = hrow - m * CF_LH;
            halo = C2[(size_t)(mc + m) * plane_stride + (size_t)row_of(r) * W + hx];
        }
    };
    auto land = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < CF_BODY; ++i) dst[(2 * i + bslot) * CF_LW + 2 + bcol] = body[i];
        if (hrow < CF_MC * CF_LH) dst[hrow * CF_LW + hlc] = halo;
    };

    float sum[4][2] = {};
    issue(0);
    land(tile[0]);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < C2N / CF_MC; ++c) {
        const float* cur = tile[c & 1];
        if (c + 1 < C2N / CF_MC) issue((c + 1) * CF_MC);
#pragma unroll 1
        for (int m = 0; m < CF_MC; ++m) {
            const float* t = cur + m * CF_CH + (wv * 4) * CF_LW + 2 * lane;     // window columns 2l-2 .. 2l+3 -> index 2l..2l+5
            const float* wm = cW.w3[c * CF_MC + m];
            float win[8][6];
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const float2 v = *reinterpret_cast<const float2*>(t + r * CF_LW + 2 * p);
                    win[r][2 * p] = v.x; win[r][2 * p + 1] = v.y;
                }
            float a[4][2] = {};
#pragma unroll
            for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) {
                    const float wgt = wm[dy * 5 + dx];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        a[q][0] = __builtin_fmaf(wgt, win[q + dy][dx], a[q][0]);
                        a[q][1] = __builtin_fmaf(wgt, win[q + dy][dx + 1], a[q][1]);
                    }
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) { sum[q][0] += a[q][0]; sum[q][1] += a[q][1]; }
        }
        if (c + 1 < C2N / CF_MC) land(tile[(c + 1) & 1]);
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = ty0 + wv * 4 + q;
        if (row < H && row < out_row0 + out_rows) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int x = tx0 + 2 * lane + e;
                if (x < W) out[(size_t)(row - out_row0) * W + x] = fminf(fmaxf(sum[q][e] + cW.b3, 0.f), 255.f);
            }
        }
    }
}

#endif  // SRCNN_STRICT_ONLY

// =============================================================================================
// Unfused layer 1 and layer 2 (stage-level entry points; not used by the hot path).
// =============================================================================================
__global__ __launch_bounds__(256) void k_conv1_planes(const float* __restrict__ Y, int W, int H,
                                                      float* __restrict__ C1, size_t plane_stride)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int k0 = blockIdx.z * 8;
    if (x >= W || y >= H) return;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 9; ++i) {
        const int gy = clampi(y + i - 4, 0, H - 1);
        for (int j = 0; j < 9; ++j) {
            const float v = Y[(size_t)gy * W + clampi(x + j - 4, 0, W - 1)];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = acc[k] + cW.w1t[i * 9 + j][k0 + k] * v;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        C1[(size_t)(k0 + k) * plane_stride + (size_t)y * W + x] = fmaxf(acc[k] + cW.b1[k0 + k], 0.f);
}

__global__ __launch_bounds__(256) void k_conv2_planes(const float* __restrict__ C1, size_t n,
                                                      float* __restrict__ C2v)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int m0 = blockIdx.y * 8;
    if (p >= n) return;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int f = 0; f < C1N; ++f) {
        const float v = C1[(size_t)f * n + p];
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m] = acc[m] + v * cW.w2[m0 + m][f];
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) C2v[(size_t)(m0 + m) * n + p] = fmaxf(acc[m] + cW.b2[m0 + m], 0.f);
}

// =============================================================================================
// Colour shell (src/libsrcnn.cpp:233-308, 889-905): u8 interleaved <-> planar float.
// =============================================================================================
__global__ __launch_bounds__(256) void k_rgb_split(const unsigned char* __restrict__ rgb, size_t n, int d,
                                                   float* __restrict__ Yp, float* __restrict__ Cb,
                                                   float* __restrict__ Cr, float* __restrict__ A)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float r = (float)rgb[p * d + 0], g = (float)rgb[p * d + 1], b = (float)rgb[p * d + 2];
    Yp[p] = (0.299f * r) + (0.587f * g) + (0.114f * b);
    Cb[p] = 128.f - (0.1687f * r) - (0.3313f * g) + (0.5f * b);
    Cr[p] = 128.f + (0.5f * r) - (0.4187f * g) - (0.0813f * b);
    if (d == 4) A[p] = (float)rgb[p * d + 3];
}

__global__ __launch_bounds__(256) void k_ycc_merge(const float* __restrict__ Yp, const float* __restrict__ Cb,
                                                   const float* __restrict__ Cr, const float* __restrict__ A,
                                                   size_t n, int d, unsigned char* __restrict__ rgb,
                                                   unsigned char* __restrict__ conv_opt)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float fy = Yp[p], cb = Cb[p] - 128.f, cr = Cr[p] - 128.f;
    rgb[p * d + 0] = to_u8_sat(fy + 45.f * cr / 32.f);
    rgb[p * d + 1] = to_u8_sat(fy - (11.f * cb + 23.f * cr) / 32.f);
    rgb[p * d + 2] = to_u8_sat(fy + 113.f * cb / 64.f);
    if (d == 4) rgb[p * d + 3] = to_u8_sat(A[p]);
    if (conv_opt) conv_opt[p] = (unsigned char)fy;
}

// =============================================================================================
// Launchers.  Shapes are validated by the C-ABI layer before any launch.
// =============================================================================================
static inline unsigned cdiv(unsigned a, unsigned b) { return (a + b - 1) / b; }

void launch_resample_cols(const float* src, int w, int src_row_base, float* dst, int dst_row0, int dst_rows,
                          const DevAxisTable& t, hipStream_t s)
{
    if (dst_rows <= 0) return;
    dim3 grid(cdiv(w, 256), std::min(dst_rows, 65535));
    hipLaunchKernelGGL(k_resample_cols, grid, dim3(256), 0, s, src, w, src_row_base, dst, dst_row0, dst_rows,
                       t.first, t.taps, t.weight, t.stride);
}

void launch_resample_rows(const float* src, int src_w, float* dst, int dst_w, int rows, const DevAxisTable& t,
                          hipStream_t s)
{
    if (rows <= 0) return;
    dim3 grid(cdiv(dst_w, 256), std::min(rows, 65535));
    hipLaunchKernelGGL(k_resample_rows, grid, dim3(256), 0, s, src, src_w, dst, dst_w, rows, t.first, t.taps,
                       t.weight, t.stride);
}

// ---- k_rs2d launchers ----
static inline bool aligned_to(const void* p, unsigned a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

namespace {
constexpr int RS_LDS_LIMIT = 96 * 1024;        // dynamic LDS a k_rs2d launch may ask for (rs2d_prepare raises the 64 KB default)

struct Rs2dPlan { int lw = 0, sr = 0, maxt = 0, tpb = 1; size_t lds = 0; unsigned gx = 0, gy = 0; bool ok = false; };

// Tile spans of one launch from the tables' host copies (the tables are monotone, so a tile's span is given by its ends).
// `sliding`: the caller wants a bound that holds for ANY alignment of the tile rows inside [dst_row0, +dst_rows) (rs2d_fits).
Rs2dPlan rs2d_plan(int np, int dst_w, int dst_row0, int dst_rows, const DevAxisTable& tv, const DevAxisTable& th, bool sliding = false)
{
    Rs2dPlan p;
    if (dst_rows <= 0 || dst_w <= 0 || !tv.monotone || !th.monotone || !tv.h_first || !th.h_first) return p;
    const int m = std::max(tv.max_taps, th.max_taps);
    if (m > 8) return p;
    p.maxt = m <= 3 ? 3 : (m <= 5 ? 5 : 8);
    int span = 0;
    for (int x0 = 0; x0 < dst_w; x0 += 256) {
        const int xl = std::min(x0 + 256, dst_w) - 1;
        span = std::max(span, th.h_first[xl] + th.h_taps[xl] - th.h_first[x0]);
    }
    // LDS row stride (compile-time in the kernel): the tile's source columns + the zero column.  136 covers every ratio >= 2.
    p.lw = span + 1 <= 136 ? 136 : (span + 1 <= 272 ? 272 : 0);
    for (int r = 0; r < dst_rows; r += (sliding ? 1 : RS_TH)) {
        const int y0 = dst_row0 + r, yl = dst_row0 + std::min(r + RS_TH, dst_rows) - 1;
        p.sr = std::max(p.sr, tv.h_first[yl] + tv.h_taps[yl] - tv.h_first[y0]);
    }
    p.lds = (size_t)rs_lds_bytes(np, p.maxt, p.sr, p.lw);
    // a block marches through `tpb` row tiles with its horizontal weights in registers; keep >= ~4 blocks per CU in the grid
    const unsigned tiles_y = cdiv(dst_rows, RS_TH);
    p.gx = cdiv(dst_w, 256);
    const int tpb_env = (int)settings().rs_tpb;
    // one round of resident blocks where possible (4 blocks per CU x 256 CUs: a grid of 1.1 rounds runs as long as one of 2:
    // measured 0.088 ms at tpb 7 = 1170 blocks vs 0.075 ms at tpb 8 = 1020 blocks for an 8K plane)
    p.tpb = tpb_env > 0 ? std::min(tpb_env, RS_MAX_TPB) : (int)std::max(1u, std::min<unsigned>(RS_MAX_TPB, cdiv(p.gx * tiles_y, 1024u)));
    p.gy = cdiv(tiles_y, p.tpb);
    p.ok = p.lw > 0 && p.sr > 0 && p.lds <= (size_t)RS_LDS_LIMIT && p.gy <= 65535u;
    return p;
}

template <int KIND, int D, bool CONV, class F>
void rs2d_variants(F&& f)
{
#define RS_ONE(MT, LW_) f(reinterpret_cast<const void*>(&k_rs2d<KIND, D, MT, LW_, CONV>), MT, LW_);
    RS_ONE(3, 136) RS_ONE(5, 136) RS_ONE(8, 136) RS_ONE(3, 272) RS_ONE(5, 272) RS_ONE(8, 272)
#undef RS_ONE
}

template <int KIND, int D, bool CONV>
void rs2d_dispatch(const Rs2dPlan& p, const Rs2dArgs& a, hipStream_t s)
{
    const dim3 grid(p.gx, p.gy), block(256);
#define RS_GO(MT, LW_) hipLaunchKernelGGL((k_rs2d<KIND, D, MT, LW_, CONV>), grid, block, p.lds, s, a)
    if (p.lw == 136) { if (p.maxt == 3) RS_GO(3, 136); else if (p.maxt == 5) RS_GO(5, 136); else RS_GO(8, 136); }
    else             { if (p.maxt == 3) RS_GO(3, 272); else if (p.maxt == 5) RS_GO(5, 272); else RS_GO(8, 272); }
#undef RS_GO
}
}  // namespace

hipError_t rs2d_prepare()
{
    hipError_t err = hipSuccess;
    auto raise = [&](const void* fn, int, int) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS_LIMIT);
        if (e != hipSuccess && err == hipSuccess) err = e;
    };
    rs2d_variants<0, 3, false>(raise);
#define RS_ONE(MT, LW_) raise(reinterpret_cast<const void*>(&k_rs2d_dma<MT, LW_>), MT, LW_);
    RS_ONE(3, 136) RS_ONE(5, 136) RS_ONE(8, 136) RS_ONE(3, 272) RS_ONE(5, 272) RS_ONE(8, 272)
#undef RS_ONE
    rs2d_variants<1, 3, false>(raise); rs2d_variants<1, 4, false>(raise);
    rs2d_variants<2, 3, false>(raise); rs2d_variants<2, 3, true>(raise);
    rs2d_variants<2, 4, false>(raise); rs2d_variants<2, 4, true>(raise);
    return err;
}

// Would k_rs2d (np = 1: plane resample; np = depth-1: fused colour merge) accept EVERY band inside rows [r0, r1)?  The bound
// is taken over all alignments of the tile rows, so a yes holds for whatever bands the caller later cuts the range into.
bool rs2d_fits(int np, int src_w, int src_h, int dst_w, int dst_h, int r0, int r1, const DevAxisTable& tv, const DevAxisTable& th)
{
    if (dst_w < src_w || dst_h < src_h || r1 <= r0) return false;
    const Rs2dPlan p = rs2d_plan(np, dst_w, r0, r1 - r0, tv, th, true);
    return p.lw > 0 && p.sr > 0 && p.maxt > 0 && p.lds <= (size_t)RS_LDS_LIMIT;
}

bool launch_rs2d(const YSource& src, int src_w, int src_h, float* dst, int dst_w, int dst_h, int dst_row0, int dst_rows,
                 const DevAxisTable& tv, const DevAxisTable& th, hipStream_t s)
{
    if (dst_w < src_w || dst_h < src_h) return false;
    if (!src.plane && !(src.rgb && (src.depth == 3 || src.depth == 4))) return false;
    const Rs2dPlan p = rs2d_plan(1, dst_w, dst_row0, dst_rows, tv, th);
    if (!p.ok) return false;
    Rs2dArgs a{};
    a.src_plane = src.plane; a.src_rgb = src.rgb; a.src_w = src_w;
    a.dst = dst; a.dst_w = dst_w; a.dst_row0 = dst_row0; a.dst_rows = dst_rows;
    a.sr = p.sr; a.tpb = p.tpb;
    a.vfirst = tv.first; a.vtaps = tv.taps; a.vwt = tv.weight; a.vstride = tv.stride;
    a.hfirst = th.first; a.htaps = th.taps; a.hwt = th.weight; a.hstride = th.stride;
    a.vec = (dst_w % 4 == 0) && aligned_to(dst, 16);
    const bool dma = settings().rs_dma;
    const size_t lds_dma = p.lds + (size_t)RS_TH * p.lw * 4;  // its intermediate rows are doubles
    if (src.plane && dma && lds_dma <= (size_t)64 * 1024) {    // (M0 carries the DMA's LDS address: stay inside what a 16-bit field reaches)
        const dim3 grid(p.gx, p.gy), block(256);
#define RS_GO(MT, LW_) hipLaunchKernelGGL((k_rs2d_dma<MT, LW_>), grid, block, lds_dma, s, a)
        if (p.lw == 136) { if (p.maxt == 3) RS_GO(3, 136); else if (p.maxt == 5) RS_GO(5, 136); else RS_GO(8, 136); }
        else             { if (p.maxt == 3) RS_GO(3, 272); else if (p.maxt == 5) RS_GO(5, 272); else RS_GO(8, 272); }
#undef RS_GO
    }
    else if (src.plane) rs2d_dispatch<0, 3, false>(p, a, s);
    else if (src.depth == 3) rs2d_dispatch<1, 3, false>(p, a, s);
    else rs2d_dispatch<1, 4, false>(p, a, s);
    return true;
}

bool launch_merge_fused(const unsigned char* rgb_src, int src_w, int src_h, int depth, const float* Yp,
                        unsigned char* rgb_out, unsigned char* conv_opt, int dst_w, int dst_h, int dst_row0, int dst_rows,
                        const DevAxisTable& tv, const DevAxisTable& th, hipStream_t s)
{
    if (dst_w < src_w || dst_h < src_h || (depth != 3 && depth != 4)) return false;
    const Rs2dPlan p = rs2d_plan(depth - 1, dst_w, dst_row0, dst_rows, tv, th);
    if (!p.ok) return false;
    Rs2dArgs a{};
    a.src_rgb = rgb_src; a.src_w = src_w;
    a.yp = Yp; a.out_rgb = rgb_out; a.out_conv = conv_opt;
    a.dst_w = dst_w; a.dst_row0 = dst_row0; a.dst_rows = dst_rows;
    a.sr = p.sr; a.tpb = p.tpb;
    a.vfirst = tv.first; a.vtaps = tv.taps; a.vwt = tv.weight; a.vstride = tv.stride;
    a.hfirst = th.first; a.htaps = th.taps; a.hwt = th.weight; a.hstride = th.stride;
    a.vec = (dst_w % 4 == 0) && aligned_to(Yp, 16) && aligned_to(rgb_out, 4) && (!conv_opt || aligned_to(conv_opt, 4));
    if (depth == 3) { if (conv_opt) rs2d_dispatch<2, 3, true>(p, a, s); else rs2d_dispatch<2, 3, false>(p, a, s); }
    else            { if (conv_opt) rs2d_dispatch<2, 4, true>(p, a, s); else rs2d_dispatch<2, 4, false>(p, a, s); }
    return true;
}

// The layer-1+2 kernel's instantiations: RELAX 0 = strict (production, and its no-DMA fallback), 3 = SRCNN_MODE_FAST, 1 / 2 = the
// single-layer relaxations of SRCNN_MODE_RELAXED (the instrument behind profiles/r04_error_matrix.txt).
template <int RELAX, bool LD>
static hipError_t prep_one()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv12_mfma<RELAX, LD>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * m_lds_floats(LD)));
}

hipError_t conv12_mfma_prepare()
{
    hipError_t e;
    if ((e = prep_one<0, true>()) != hipSuccess) return e;
    if ((e = prep_one<0, false>()) != hipSuccess) return e;
#ifndef SRCNN_STRICT_ONLY            // make STRICT_ONLY=1: no instance of a non-parity kernel is compiled
    if ((e = prep_one<1, true>()) != hipSuccess) return e;
    if ((e = prep_one<2, true>()) != hipSuccess) return e;
    if ((e = prep_one<3, true>()) != hipSuccess) return e;
#endif
    return hipSuccess;
}

// Resident workgroups of the layer-1+2 kernel (it loops over 64 x 16 tiles) and its tile height: what a band planner needs
// to cut bands whose tile count fills whole rounds of the grid.
void conv12_grid_info(int num_cus, int* blocks, int* tile_rows)
{
    if (blocks) *blocks = M_BPC * num_cus;
    if (tile_rows) *tile_rows = M_TH;
}

void launch_conv12_mfma(const float* Y, int W, int H, int y_row_base, int y_rows, float* C2, size_t plane_stride, int out_row0,
                        int out_rows, int relax, int num_cus, hipStream_t s, unsigned long long* clk, unsigned* queue)
{
    if (out_rows <= 0) return;
    const int tiles_x = (int)cdiv(W, M_TW), tiles_y = (int)cdiv(out_rows, M_TH);
    const int ntiles = tiles_x * tiles_y;
    // resident blocks only; tile loop inside.  Fewer tiles than resident blocks (small images, short bands): the kernel deals
    // such a "last round" out in quarter tiles, so up to four blocks share a tile instead of three quarters of the chip idling
    const int cap = M_BPC * num_cus;
    const int grid = ntiles >= cap ? cap : (settings().conv12_spread ? std::min(4 * ntiles, cap) : ntiles);
#define CONV12_GO(R, LD) hipLaunchKernelGGL((k_conv12_mfma<R, LD>), dim3(grid), dim3(64 * M_NW), sizeof(float) * m_lds_floats(LD), s, Y, W, H, \
                                           y_row_base, y_rows, C2, plane_stride, out_row0, out_rows, tiles_x, ntiles, clk, LD ? queue : nullptr)
#ifdef SRCNN_STRICT_ONLY
    (void)relax;                     // srcnn_set_mode refuses every other mode in this build
    if (settings().conv12_dma) CONV12_GO(0, true); else CONV12_GO(0, false);
#else
    switch (relax & 3) {
    case 0: if (settings().conv12_dma) CONV12_GO(0, true); else CONV12_GO(0, false); break;
    case 1: CONV12_GO(1, true); break;
    case 2: CONV12_GO(2, true); break;
    default: CONV12_GO(3, true); break;
    }
#endif
#undef CONV12_GO
}

void launch_conv3(const float* C2, size_t plane_stride, int W, int H, int c2_row_base, int c2_rows, float* out,
                  int out_row0, int out_rows, int relax, hipStream_t s)
{
    if (out_rows <= 0) return;
#ifndef SRCNN_STRICT_ONLY
    const bool strict = !(relax & (RELAX_L3_X64 | RELAX_L3_F32));
    const bool wide_planes = (size_t)c2_rows * (size_t)W * sizeof(float) >= ((size_t)1 << 32);
    if (relax & RELAX_L3_X64) {
        const dim3 gx(cdiv(W, 64), cdiv(out_rows, 16));
        if (wide_planes)
            hipLaunchKernelGGL((k_conv3<true, true, true>), gx, dim3(256), 0, s, C2, plane_stride, W, H, c2_row_base, c2_rows, out, out_row0, out_rows, 0);
        else
            hipLaunchKernelGGL((k_conv3<true, false, true>), gx, dim3(256), 0, s, C2, plane_stride, W, H, c2_row_base, c2_rows, out, out_row0, out_rows, 0);
        return;
    }
    if (!strict) {
        dim3 gridf(cdiv(W, CF_TW), cdiv(out_rows, CF_TH));
        hipLaunchKernelGGL(k_conv3_fast, gridf, dim3(256), 0, s, C2, plane_stride, W, H, c2_row_base, c2_rows, out,
                           out_row0, out_rows);
        return;
    }
#else
    (void)relax;
#endif
    dim3 grid(cdiv(W, 64), cdiv(out_rows, 16));
    const bool force_wide = settings().conv3_off64;      // test hook
    const int wdma = settings().conv3_wdma ? 1 : 0;
    const bool wide = force_wide || (size_t)c2_rows * (size_t)W * sizeof(float) >= ((size_t)1 << 32);     // per-plane byte offsets beyond 32 bits
    if (!wide && wdma)
        hipLaunchKernelGGL((k_conv3<true, false, false, true>), grid, dim3(256), 0, s, C2, plane_stride, W, H, c2_row_base, c2_rows,
                           out, out_row0, out_rows, 1);
    else if (!wide)
        hipLaunchKernelGGL((k_conv3<true, false>), grid, dim3(256), 0, s, C2, plane_stride, W, H, c2_row_base, c2_rows,
                           out, out_row0, out_rows, wdma);
    else
        hipLaunchKernelGGL((k_conv3<true, true>), grid, dim3(256), 0, s, C2, plane_stride, W, H, c2_row_base, c2_rows,
                           out, out_row0, out_rows, wdma);
}

void launch_conv1_planes(const float* Y, int W, int H, float* C1, hipStream_t s)
{
    dim3 grid(cdiv(W, 64), cdiv(H, 4), C1N / 8);
    hipLaunchKernelGGL(k_conv1_planes, grid, dim3(256), 0, s, Y, W, H, C1, (size_t)W * H);
}

void launch_conv2_planes(const float* C1, size_t n, float* C2v, hipStream_t s)
{
    dim3 grid((unsigned)((n + 255) / 256), C2N / 8);
    hipLaunchKernelGGL(k_conv2_planes, grid, dim3(256), 0, s, C1, n, C2v);
}

// ---- 4-pixels-per-thread forms of the colour shell: 16-byte plane accesses, 12/16-byte packed pixel
//      accesses.  Same per-pixel arithmetic as the scalar kernels (which remain for unaligned spans/tails). ----
template <int D>
__global__ __launch_bounds__(256) void k_rgb_split4(const unsigned* __restrict__ rgb32, size_t n4,
                                                    float4* __restrict__ Yp, float4* __restrict__ Cb,
                                                    float4* __restrict__ Cr, float4* __restrict__ A)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n4) return;
    unsigned wds[D];
#pragma unroll
    for (int k = 0; k < D; ++k) wds[k] = rgb32[q * D + k];            // D dwords = 4 pixels of D bytes
    float yv[4], cbv[4], crv[4], av[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float ch[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int byte = i * D + k;
            ch[k] = (float)((wds[byte >> 2] >> (8 * (byte & 3))) & 0xffu);
        }
        const float r = ch[0], g = ch[1], b = ch[2];
        yv[i] = (0.299f * r) + (0.587f * g) + (0.114f * b);
        cbv[i] = 128.f - (0.1687f * r) - (0.3313f * g) + (0.5f * b);
        crv[i] = 128.f + (0.5f * r) - (0.4187f * g) - (0.0813f * b);
        av[i] = ch[3];
    }
    Yp[q] = make_float4(yv[0], yv[1], yv[2], yv[3]);
    Cb[q] = make_float4(cbv[0], cbv[1], cbv[2], cbv[3]);
    Cr[q] = make_float4(crv[0], crv[1], crv[2], crv[3]);
    if (D == 4) A[q] = make_float4(av[0], av[1], av[2], av[3]);
}

template <int D, bool CONV>
__global__ __launch_bounds__(256) void k_ycc_merge4(const float4* __restrict__ Yp, const float4* __restrict__ Cb,
                                                    const float4* __restrict__ Cr, const float4* __restrict__ A,
                                                    size_t n4, unsigned* __restrict__ rgb32, unsigned* __restrict__ conv32)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n4) return;
    const float4 y4 = Yp[q], cb4 = Cb[q], cr4 = Cr[q];
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (D == 4) a4 = A[q];
    const float ys[4] = {y4.x, y4.y, y4.z, y4.w}, cbs[4] = {cb4.x, cb4.y, cb4.z, cb4.w};
    const float crs[4] = {cr4.x, cr4.y, cr4.z, cr4.w}, as[4] = {a4.x, a4.y, a4.z, a4.w};
    unsigned wds[D] = {};
    unsigned cw = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float fy = ys[i], cb = cbs[i] - 128.f, cr = crs[i] - 128.f;
        unsigned char px[4];
        px[0] = to_u8_sat(fy + 45.f * cr / 32.f);
        px[1] = to_u8_sat(fy - (11.f * cb + 23.f * cr) / 32.f);
        px[2] = to_u8_sat(fy + 113.f * cb / 64.f);
        px[3] = to_u8_sat(as[i]);
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int byte = i * D + k;
            wds[byte >> 2] |= (unsigned)px[k] << (8 * (byte & 3));
        }
        cw |= (unsigned)(unsigned char)fy << (8 * i);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) rgb32[q * D + k] = wds[k];
    if (CONV) conv32[q] = cw;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

void launch_rgb_split(const unsigned char* rgb, size_t n, int d, float* Yp, float* Cb, float* Cr, float* A,
                      hipStream_t s)
{
    size_t done = 0;
    if ((d == 3 || d == 4) && n >= 1024 && aligned16(Yp) && aligned16(Cb) && aligned16(Cr) && (d == 3 || aligned16(A)) &&
        (reinterpret_cast<uintptr_t>(rgb) & 3u) == 0) {
        const size_t n4 = n / 4;
        const dim3 grid((unsigned)((n4 + 255) / 256));
        if (d == 3)
            hipLaunchKernelGGL((k_rgb_split4<3>), grid, dim3(256), 0, s, reinterpret_cast<const unsigned*>(rgb), n4,
                               reinterpret_cast<float4*>(Yp), reinterpret_cast<float4*>(Cb), reinterpret_cast<float4*>(Cr),
                               reinterpret_cast<float4*>(A));
        else
            hipLaunchKernelGGL((k_rgb_split4<4>), grid, dim3(256), 0, s, reinterpret_cast<const unsigned*>(rgb), n4,
                               reinterpret_cast<float4*>(Yp), reinterpret_cast<float4*>(Cb), reinterpret_cast<float4*>(Cr),
                               reinterpret_cast<float4*>(A));
        done = n4 * 4;
    }
    if (done < n) {
        const size_t rem = n - done;
        hipLaunchKernelGGL(k_rgb_split, dim3((unsigned)((rem + 255) / 256)), dim3(256), 0, s, rgb + done * d, rem, d,
                           Yp + done, Cb + done, Cr + done, A ? A + done : A);
    }
}

void launch_ycc_merge(const float* Yp, const float* Cb, const float* Cr, const float* A, size_t n, int d,
                      unsigned char* rgb, unsigned char* conv_opt, hipStream_t s)
{
    size_t done = 0;
    if ((d == 3 || d == 4) && n >= 1024 && aligned16(Yp) && aligned16(Cb) && aligned16(Cr) && (d == 3 || aligned16(A)) &&
        (reinterpret_cast<uintptr_t>(rgb) & 3u) == 0 && (!conv_opt || (reinterpret_cast<uintptr_t>(conv_opt) & 3u) == 0)) {
        const size_t n4 = n / 4;
        const dim3 grid((unsigned)((n4 + 255) / 256));
        const float4 *y4 = reinterpret_cast<const float4*>(Yp), *cb4 = reinterpret_cast<const float4*>(Cb);
        const float4 *cr4 = reinterpret_cast<const float4*>(Cr), *a4 = reinterpret_cast<const float4*>(A);
        unsigned* o32 = reinterpret_cast<unsigned*>(rgb);
        unsigned* c32 = reinterpret_cast<unsigned*>(conv_opt);
        if (d == 3 && conv_opt) hipLaunchKernelGGL((k_ycc_merge4<3, true>), grid, dim3(256), 0, s, y4, cb4, cr4, a4, n4, o32, c32);
        else if (d == 3) hipLaunchKernelGGL((k_ycc_merge4<3, false>), grid, dim3(256), 0, s, y4, cb4, cr4, a4, n4, o32, c32);
        else if (conv_opt) hipLaunchKernelGGL((k_ycc_merge4<4, true>), grid, dim3(256), 0, s, y4, cb4, cr4, a4, n4, o32, c32);
        else hipLaunchKernelGGL((k_ycc_merge4<4, false>), grid, dim3(256), 0, s, y4, cb4, cr4, a4, n4, o32, c32);
        done = n4 * 4;
    }
    if (done < n) {
        const size_t rem = n - done;
        hipLaunchKernelGGL(k_ycc_merge, dim3((unsigned)((rem + 255) / 256)), dim3(256), 0, s, Yp + done, Cb + done, Cr + done,
                           A ? A + done : A, rem, d, rgb + done * d, conv_opt ? conv_opt + done : conv_opt);
    }
}

}  // namespace srcnn
