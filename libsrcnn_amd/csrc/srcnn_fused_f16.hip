// srcnn_fused_f16.hip -- the whole convolution stack as ONE gfx950 kernel for the non-parity tier
// SRCNN_MODE_FAST_F16: upscaled Y in, Y' out, nothing in between ever reaches HBM.
//
// Reference behaviour approximated (never bit-exact; bound and measurement in tests/ and bench.py):
//   64 x convolution99 + ReLU      src/libsrcnn.cpp:350-422, :785-798
//   32 x convolution11 + ReLU      src/libsrcnn.cpp:424-447, :811-824
//   convolution55 + clamp          src/libsrcnn.cpp:449-529, :838-846
//
// Arithmetic: every fp32 operand is split into two fp16 pieces (x = hi + lo to 22 bits; weights pre-scaled by 2^8
// so their low pieces stay out of the fp16 subnormal range) and every product is hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation -- the error class of an fp32 FMA evaluation.
//
// Layer 3 has ONE output channel, which would waste 31/32 of a matrix instruction.  It is therefore evaluated as a
// 1x1 convolution with 25 outputs -- P[tap][px] = sum_m w3[m][tap] * c2[m][px], a 32(25 used) x 32 x 32 GEMM whose B
// operand is the layer-2 accumulator tile itself -- followed by a 25-term shifted sum
//     Y'[y][x] = b3 + sum_{dy,dx} P[dy,dx][clamp(y+dy-2)][clamp(x+dx-2)]
// (clamp-to-edge of the ACTIVATIONS is clamp-to-edge of P).  So the 800 MACs per pixel of layer 3 become 6 MFMAs per
// 32 pixels plus 25 LDS reads and fp32 adds per pixel.
//
// Decomposition.  A workgroup of NW waves owns a strip of NW*60 output columns and marches down a chunk of rows.
// Each WAVE owns 64 layer-2 columns (two 32-pixel MFMA segments) = 60 output columns + the 2+2 columns of layer-3
// halo, recomputed per wave (6.7 %) so that waves never exchange activations: a wave writes the P rows of its own
// 64 columns to its private LDS buffer and gathers from it.  Vertically nothing is recomputed inside a chunk: the
// five partial output rows that a layer-2 row contributes to live in registers O[0..4] of the lane that owns the
// output column and rotate as the wave moves down; a row is stored when its fifth contribution has arrived.
// Chunk starts cost 4 warm-up rows.  Each wave also stages its OWN 72 columns of upscaled Y through a private 12-row LDS
// ring (a 4-row stage + 4 halo rows either side), fetched one stage ahead into registers and stored as fp16 hi and lo
// planes, each in two copies shifted by one element: a lane's B fragment is 8 consecutive halves starting at an
// arbitrary column, and with the copy picked by the parity of that column it is always 4-byte aligned (two
// ds_read2_b32, no repacking).  After the weights are in LDS the waves of a workgroup never synchronise again: there is
// no barrier in the main loop, so the two waves that share a SIMD can sit in different phases -- one in the MFMA-only
// layer-1 phase while the other does the VALU-heavy layer-2/3 hand-offs -- instead of marching in lockstep.
//
// LDS (NW = 8): weight fragments 48 KB + Y rings 60 KB + P buffers 50 KB = 158 KB -> one workgroup per CU, two waves
// per SIMD.  HBM traffic: 4 B in + 4 B out per pixel (+ 6.7 % / chunk-halo re-reads, all L2 hits).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "srcnn_kernels.h"

#pragma clang fp contract(off)

namespace srcnn {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NW = FU_NW;                        // waves per workgroup
constexpr int NT = 64 * NW;
constexpr int OW = 60;                           // output columns per wave
constexpr int GW = NW * OW;                      // output columns per workgroup
constexpr int TWW = 72;                          // staged Y columns per wave: 64 layer-2 columns +-4
constexpr int PB = 160;                          // bytes per ring plane row: 80 halves >= TWW + 1 (shifted copy) + 7
constexpr int SB = 4 * PB;                       // bytes per ring slot: planes hi0, hi1 (shifted), lo0, lo1 (shifted)
constexpr int SLOTS = 12, STAGE = 4;             // ring rows / rows per stage (SLOTS = STAGE + 8 halo rows)
constexpr int RING = SLOTS * SB;                 // bytes per wave
constexpr int PS = 64;                           // P plane stride (floats)
constexpr int PW = 25 * PS;                      // floats per wave
constexpr float F_INV = 1.f / 256.f;             // undoes FusedF16Weights' 2^8 weight scale (exact)

constexpr int L_W1 = 0;                                          // byte offsets into dynamic LDS
constexpr int L_W2 = L_W1 + (int)sizeof(FusedF16Weights::w1);
constexpr int L_W3 = L_W2 + (int)sizeof(FusedF16Weights::w2);
constexpr int L_Y = L_W3 + (int)sizeof(FusedF16Weights::w3);
constexpr int L_P = L_Y + NW * RING;
constexpr int L_END = L_P + NW * PW * 4;
static_assert(L_Y % 16 == 0 && L_P % 16 == 0, "alignment");
static_assert(L_END <= 160 * 1024, "LDS budget");
static_assert(offsetof(FusedF16Weights, b1) == L_Y, "the weight fragments are the head of the blob");

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// x = hi + lo with hi, lo in fp16 (22 significant bits together); the residual x - hi is exact in fp32 and is written
// as an fma so that it becomes one v_fma_mix_f32 reading the fp16 register directly
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)v;
    lo = (_Float16)__builtin_fmaf((float)hi, -1.0f, v);
}

// 8 accumulator values -> the hi and lo B fragments of one 16-deep k-step.  Two values at a time: pack their fp16 heads
// (round to nearest), subtract each head from its value with v_fma_mix_f32 reading the fp16 half in place (exact), pack
// the fp16 tails: 4 instructions per pair.  (Left to the compiler the same arithmetic takes ~7: it unpacks the heads
// back to fp32 first.)
__device__ __forceinline__ void split8(const float* x, h8& xh, h8& xl)
{
    u32x4 hv, lv;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned hp, lp;
        float r0, r1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(x[2 * p]), "v"(x[2 * p + 1]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hp), "v"(x[2 * p]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hp), "v"(x[2 * p + 1]));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lp) : "v"(r0), "v"(r1));
        hv[p] = hp; lv[p] = lp;
    }
    xh = __builtin_bit_cast(h8, hv);
    xl = __builtin_bit_cast(h8, lv);
}

__device__ __forceinline__ h8 as_h8(u32x4 v) { return __builtin_bit_cast(h8, v); }

__device__ __forceinline__ int mod12(int v) { return v - 12 * (v / 12); }      // v >= 0

}  // namespace

// DIAG builds (srcnn_fused_diag, tools only) stamp s_memtime around the phases of every row of workgroup 0 into `dbg`;
// the production instantiation has no stamp and no extra argument use.
template <bool DIAG>
__global__ __launch_bounds__(NT) void k_fused_f16(
    const float* __restrict__ Y, int W, int H, int y_row_base, int y_rows,     // Y holds rows [y_row_base, +y_rows)
    float* __restrict__ out, int out_row0, int out_rows,                        // writes rows [out_row0, +out_rows)
    const FusedF16Weights* __restrict__ blob, int chunk_rows, int tiles_x, int skew, unsigned long long* __restrict__ dbg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const _Float16* W1f = reinterpret_cast<const _Float16*>(lds_raw + L_W1);
    const _Float16* W2f = reinterpret_cast<const _Float16*>(lds_raw + L_W2);
    const _Float16* W3f = reinterpret_cast<const _Float16*>(lds_raw + L_W3);
    const float* B1s = blob->b1;                                // biases: read once, live in registers
    const float* B2s = blob->b2;
    float* Pall = reinterpret_cast<float*>(lds_raw + L_P);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5, col = lane & 31;
    const int txi = blockIdx.x % tiles_x, cyi = blockIdx.x / tiles_x;
    const int GX0 = txi * GW;                                   // first output column of the workgroup
    const int R0 = out_row0 + cyi * chunk_rows;                 // output rows [R0, R1) of this chunk
    const int R1 = min(R0 + chunk_rows, out_row0 + out_rows);
    if (R0 >= R1) return;                                       // uniform per workgroup, before the barrier

    {   // weight fragments: the head of the blob is the LDS image
        const uint4* src = reinterpret_cast<const uint4*>(blob);
        uint4* dst = reinterpret_cast<uint4*>(lds_raw);
        for (int e = tid; e < L_Y / 16; e += NT) dst[e] = src[e];
    }
    const float b3 = blob->b3;
    __syncthreads();                                            // the only workgroup barrier of the kernel

    const int cx0 = GX0 + OW * wv - 2;                          // image column of this wave's layer-2 column 0
    const int TX0 = cx0 - 4;                                    // image column of this wave's staged column 0
    const int ubase = R0 - 6;                                   // virtual Y row held by ring slot 0 (mod 12)
    const int y_last = y_row_base + y_rows - 1;
    unsigned char* Yr = lds_raw + L_Y + wv * RING;              // this wave's ring: [slot][hi0 | hi1 | lo0 | lo1][80 halves]

    // ---- Y staging, per wave: rows are "virtual" (may lie outside the image; clamped when fetched) ----
    constexpr int PRE = (STAGE * TWW + 63) / 64;
    float pre[PRE];
    auto fetch = [&](int u0) {          // virtual rows [u0, u0 + STAGE) -> registers
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const int e = lane + 64 * k;
            const int r = e / TWW, t = e - r * TWW;
            const int gy = clampi(clampi(u0 + r, 0, H - 1), y_row_base, y_last) - y_row_base;
            const int gx = clampi(TX0 + t, 0, W - 1);
            pre[k] = (e < STAGE * TWW) ? Y[(size_t)gy * W + gx] : 0.f;
        }
    };
    auto land = [&](int u0) {           // registers -> ring slots of rows [u0, u0 + STAGE); u0 - ubase is a multiple of STAGE
        const int slot0 = mod12(u0 - ubase);
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const int e = lane + 64 * k;
            if (e < STAGE * TWW) {
                const int r = e / TWW, t = e - r * TWW;
                _Float16 hi, lo;
                split_f16(pre[k], hi, lo);
                _Float16* row = reinterpret_cast<_Float16*>(Yr + (slot0 + r) * SB);
                row[t] = hi;                      row[PB / 2 + t + 1] = hi;          // copy 1 holds column t at t + 1
                row[PB + t] = lo;                 row[3 * PB / 2 + t + 1] = lo;
            }
        }
    };
    auto wave_sync = [] {               // LDS hand-over between the lanes of ONE wave: no s_barrier involved
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int k = 0; k < SLOTS / STAGE; ++k) { fetch(ubase + STAGE * k); land(ubase + STAGE * k); }
    wave_sync();
    // Waves w and w + NW/2 share a SIMD.  A one-off offset of about half a row puts them in opposite phases.
    if (wv >= NW / 2)
        for (int i = 0; i < skew; ++i) __builtin_amdgcn_s_sleep(2);
    if constexpr (DIAG) { if (skew < 0 && wv >= NW / 2) return; }      // timing experiment: one wave per SIMD (output incomplete)

    // ---- per-lane constants ----
    float* Pw = Pall + wv * PW;
    int pidx[5];                                                // P column read for dx = 0..4 (clamped to the image)
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) pidx[dx] = clampi(clampi(cx0 + lane + dx - 2, 0, W - 1) - cx0, 0, 63);
    const int ox = cx0 + lane;                                  // output column of this lane
    const bool ox_ok = lane >= 2 && lane < 2 + OW && ox < W;
    // LDS byte address (inside a ring slot) of this lane's B fragment for segment 0: staged column q0, from the copy
    // whose alignment suits its parity; segment 1 is +64 B, the lo plane +2*PB.  Kept opaque so that every read below
    // is "one base register + immediate offset".
    const int q0 = col + half;
    unsigned frag_hi = (unsigned)(L_Y + wv * RING + ((q0 & 1) ? PB + 2 * (q0 + 1) : 2 * q0));
    asm volatile("" : "+v"(frag_hi));

    float O[5] = {0.f, 0.f, 0.f, 0.f, 0.f};                     // partial sums of output rows v+2 .. v-2
    int a_prev = -0x40000000;
    const int nv = R1 - R0 + 4;                                 // virtual layer-2 rows R0-2 .. R1+1

    for (int st = 0; st * STAGE < nv; ++st) {
        const bool more = (st + 1) * STAGE < nv;
        if (more) fetch(ubase + SLOTS + STAGE * st);
#pragma unroll 1
        for (int i = 0; i < STAGE; ++i) {
            const int v = R0 - 2 + STAGE * st + i;
            if (v > R1 + 1) break;
            const int a = clampi(v, 0, H - 1);                  // the reference clamps layer-2 ACTIVATIONS at the border
            unsigned long long t_a = 0, t_b = 0, t_c = 0;
            if constexpr (DIAG) t_a = __builtin_amdgcn_s_memtime();
            if (a != a_prev) {
                a_prev = a;
                // ================= layer 1: 9 k-steps (one window row each), both segments share the A fragments =====
                f32x16 acc[2][2] = {};
                const int s0 = mod12(a - 4 - ubase);               // ring slot of the first window row (uniform)
                // Software pipeline: the fragments of k-step s+1 are requested from LDS BEFORE the 12 MFMAs of k-step s
                // are issued, so their latency hides under 384 cycles of matrix work instead of stalling the wave.
                h8 bh[2], bl[2], a0h, a0l, a1h, a1l;
                auto load_step = [&](int s, h8 (&xbh)[2], h8 (&xbl)[2], h8& x0h, h8& x0l, h8& x1h, h8& x1l) {
                    const int slot = s0 + s >= SLOTS ? s0 + s - SLOTS : s0 + s;
                    const unsigned* yh = reinterpret_cast<const unsigned*>(lds_raw + (frag_hi + slot * SB));
                    const unsigned* yl = yh + 2 * PB / 4;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        u32x4 hi4, lo4;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { hi4[q] = yh[16 * g + q]; lo4[q] = yl[16 * g + q]; }
                        xbh[g] = as_h8(hi4); xbl[g] = as_h8(lo4);
                    }
                    x0h = *reinterpret_cast<const h8*>(W1f + (((s * 2 + 0) * 2 + 0) * 64 + lane) * 8);
                    x0l = *reinterpret_cast<const h8*>(W1f + (((s * 2 + 0) * 2 + 1) * 64 + lane) * 8);
                    x1h = *reinterpret_cast<const h8*>(W1f + (((s * 2 + 1) * 2 + 0) * 64 + lane) * 8);
                    x1l = *reinterpret_cast<const h8*>(W1f + (((s * 2 + 1) * 2 + 1) * 64 + lane) * 8);
                };
                load_step(0, bh, bl, a0h, a0l, a1h, a1l);
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    h8 nbh[2], nbl[2], n0h, n0l, n1h, n1l;
                    if (s + 1 < 9) load_step(s + 1, nbh, nbl, n0h, n0l, n1h, n1l);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bh[g], acc[g][0], 0, 0, 0);
                        acc[g][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bh[g], acc[g][1], 0, 0, 0);
                        acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bl[g], acc[g][0], 0, 0, 0);
                        acc[g][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bl[g], acc[g][1], 0, 0, 0);
                        acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, bh[g], acc[g][0], 0, 0, 0);
                        acc[g][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, bh[g], acc[g][1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < 9) {
                        bh[0] = nbh[0]; bh[1] = nbh[1]; bl[0] = nbl[0]; bl[1] = nbl[1];
                        a0h = n0h; a0l = n0l; a1h = n1h; a1l = n1l;
                    }
                }
                if constexpr (DIAG) { asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[1][1][15])); t_b = __builtin_amdgcn_s_memtime(); }
                // ================= layers 2 and 3 per segment: accumulator tiles are the next B operands =============
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    f32x16 acc2 = {};
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            float x[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                x[j] = fmaxf(__builtin_fmaf(acc[g][blk][8 * ks + j], F_INV, B1s[half * 32 + 16 * blk + 8 * ks + j]), 0.f);
                            h8 xh, xl;
                            split8(x, xh, xl);
                            const h8 ah = *reinterpret_cast<const h8*>(W2f + (((blk * 2 + ks) * 2 + 0) * 64 + lane) * 8);
                            const h8 al = *reinterpret_cast<const h8*>(W2f + (((blk * 2 + ks) * 2 + 1) * 64 + lane) * 8);
                            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, acc2, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl, acc2, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh, acc2, 0, 0, 0);
                        }
                    f32x16 accp = {};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        float x[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            x[j] = fmaxf(__builtin_fmaf(acc2[8 * ks + j], F_INV, B2s[half * 16 + 8 * ks + j]), 0.f);
                        h8 xh, xl;
                        split8(x, xh, xl);
                        const h8 ah = *reinterpret_cast<const h8*>(W3f + ((ks * 2 + 0) * 64 + lane) * 8);
                        const h8 al = *reinterpret_cast<const h8*>(W3f + ((ks * 2 + 1) * 64 + lane) * 8);
                        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh, accp, 0, 0, 0);
                        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl, accp, 0, 0, 0);
                        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh, accp, 0, 0, 0);
                    }
                    // P rows of this segment: register r of half h is tap 8*(r/4) + 4h + r%4; taps >= 25 are padding
#pragma unroll
                    for (int r = 0; r < 13; ++r) {
                        const int t0 = 8 * (r >> 2) + (r & 3);          // + 4*half
                        if (r < 12 || half == 0) Pw[(t0 + 4 * half) * PS + 32 * g + col] = accp[r] * F_INV;
                    }
                }
                wave_sync();
            }
            if constexpr (DIAG) t_c = __builtin_amdgcn_s_memtime();
            // ================= layer-3 gather: layer-2 row v feeds output rows v+2-dy with tap row dy ===================
#pragma unroll
            for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) O[dy] += Pw[(dy * 5 + dx) * PS + pidx[dx]];
            const int orow = v - 2;                                 // complete: its last contribution was tap row 4
            if (orow >= R0 && ox_ok) out[(size_t)(orow - out_row0) * W + ox] = fminf(fmaxf(O[4] + b3, 0.f), 255.f);
            O[4] = O[3]; O[3] = O[2]; O[2] = O[1]; O[1] = O[0]; O[0] = 0.f;
            if constexpr (DIAG) {
                const int idx = STAGE * st + i;
                if (blockIdx.x == 0 && lane == 0 && idx < 64 && dbg) {
                    unsigned long long* d = dbg + ((size_t)wv * 64 + idx) * 4;
                    d[0] = t_a; d[1] = t_b; d[2] = t_c; d[3] = __builtin_amdgcn_s_memtime();
                }
            }
            __builtin_amdgcn_wave_barrier();                        // the next row overwrites Pw
        }
        if (more) {
            wave_sync();                                        // this wave has finished reading the slots it overwrites
            land(ubase + SLOTS + STAGE * st);
            wave_sync();
        }
    }
}

hipError_t fused_f16_prepare()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_f16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, L_END);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_f16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, L_END);
}

void launch_fused_f16(const float* Y, int W, int H, int y_row_base, int y_rows, float* out, int out_row0, int out_rows,
                      const FusedF16Weights* d_blob, int num_cus, int skew, hipStream_t s, unsigned long long* dbg)
{
    if (out_rows <= 0) return;
    const int tiles_x = (W + GW - 1) / GW;
    // one workgroup per CU: cut the rows into just enough chunks to fill the chip (a chunk start costs 4 warm-up rows)
    int chunks = std::max(1, (num_cus + tiles_x - 1) / tiles_x);
    chunks = std::min(chunks, std::max(1, out_rows / 16));
    const int chunk_rows = (out_rows + chunks - 1) / chunks;
    chunks = (out_rows + chunk_rows - 1) / chunk_rows;
    if (dbg)
        hipLaunchKernelGGL(k_fused_f16<true>, dim3(tiles_x * chunks), dim3(NT), L_END, s, Y, W, H, y_row_base, y_rows, out, out_row0,
                           out_rows, d_blob, chunk_rows, tiles_x, skew, dbg);
    else
        hipLaunchKernelGGL(k_fused_f16<false>, dim3(tiles_x * chunks), dim3(NT), L_END, s, Y, W, H, y_row_base, y_rows, out, out_row0,
                           out_rows, d_blob, chunk_rows, tiles_x, skew, dbg);
}

}  // namespace srcnn
