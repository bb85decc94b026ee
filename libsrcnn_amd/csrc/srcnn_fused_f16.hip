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
// Layer 1 is a [64 x 81] x [81 x px] GEMM.  80 of its 81 taps are packed into 5 k-steps of 16 slots (not 9 x 16 = 144):
// four k-steps take two window rows each (lanes 0-31 one row, lanes 32-63 the next; taps dx 0..7 = 8 consecutive
// pixels), the fifth takes window row 8 row-wise and the tap column dx = 8 of rows 0..7 column-wise (16-bit LDS reads
// with one address per element).  The 81st tap (8,8) is an exact fp32 FMA into the C operand that opens the chains,
// next to the bias.  60 + 36 = 96 MFMAs per 2 x 32 pixels.
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
// Round 3: the slot layout is [hi0 | lo0 | hi1 | lo1] with copy 1 shifted by -1, which puts the even lanes' fragment reads on
// banks 0-15 and the odd lanes' on banks 16-31 of a ds_read_b32 lane group; round 2's [hi0 | hi1 | lo0 | lo1] with a +1 shift
// had them overlap on 7 banks, a 2-way conflict on every layer-1 fragment read (SQ_LDS_BANK_CONFLICT 7.4e7 -> 1.7e6 cycles
// per 8K frame, 20 % -> 0.6 % of the LDS-active cycles; kernel 1.53 -> 1.42 ms; profiles/r03_fused_bank_conflicts.txt).
//
// LDS (NW = 8): weight fragments + biases 48.4 KB + Y rings 60 KB + P buffers 50 KB = 158.4 KB -> one workgroup per
// CU, two waves per SIMD.  HBM traffic: 4 B in + 4 B out per pixel (+ 6.7 % / chunk-halo re-reads, all L2 hits).
//
// Schedule (measured on MI355X with tools/fused_timeline.py, 7680x4320 frame; profiles/r02_fused_variants.txt):
//   * Every LDS operand -- layer-1 fragments, W2/W3 fragments, biases -- is requested one 12-MFMA region before its use,
//     pinned with sched_barrier: left alone the compiler re-loads operands right before they are needed (~70 exposed LDS
//     latencies per row).  Biases enter as the C operand of the MFMA that opens a chain, not as VALU adds.
//   * The hi/lo split is 4 instructions per pair of values; only v_fma_mix_f32 is inline asm, both fp16 conversions are
//     compiler-visible so that every register an MFMA reads was written by an instruction the hazard recogniser sees
//     (asm feeding an MFMA directly produced wrong results as soon as the scheduler moved the MFMA next to it).
//   * FU_SEQ=1 (default): per row, layer 1 (60 MFMAs + the 64 FMAs of the 81st tap) then layers 2+3 and the gather as one region (36
//     MFMAs + all the VALU work, both segments' chains interleaved), with a higher wave priority in the second phase so
//     that the two waves of a SIMD complement each other.  Without priorities the older wave wins every arbitration
//     (9.5k vs 16.7k cycles per row, the workgroup waits for the slow half).  Per row and wave: layer 1 ~4.9-6.0k
//     cycles, layers 2+3 + gather ~4.0k; the matrix pipe is busy 63-73 % of the time.
//   * FU_SEQ=0: the previous row's layers 2+3 cut into nine slices and woven into the next row's layer-1 k-steps
//     (sched_group_barrier).  Needs both accumulator sets: 256 VGPRs spill at two waves per SIMD; with one wave per
//     SIMD (FU_NW_DEF=4, 428 registers) it runs 7.6k cycles per row -- slower than two simpler waves.  Kept for A/B.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>
#include "srcnn_kernels.h"

#pragma clang fp contract(off)

namespace srcnn {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NW = FU_NW;                        // waves per workgroup
constexpr int NT = 64 * NW;
#ifndef FU_SEG
#define FU_SEG 2
#endif
constexpr int SEG = FU_SEG;                      // 32-pixel MFMA segments per wave (layer-2 columns = 32 * SEG)
constexpr int OW = 32 * SEG - 4;                 // output columns per wave
constexpr int GW = NW * OW;                      // output columns per workgroup
constexpr int TWW = 32 * SEG + 8;                // staged Y columns per wave: the layer-2 columns +-4
constexpr int PB = 2 * (TWW + 8);                // bytes per ring plane row: TWW + 1 (shifted copy) + 7 halves
constexpr int SB = 4 * PB;                       // bytes per ring slot: planes hi0, lo0, hi1 (shifted), lo1 (shifted)
constexpr int SLOTS = 12, STAGE = 4;             // ring rows / rows per stage (SLOTS = STAGE + 8 halo rows)
constexpr int RING = SLOTS * SB;                 // bytes per wave
constexpr int PS = 32 * SEG;                     // P plane stride (floats)
constexpr int PW = 25 * PS;                      // floats per wave
constexpr int NK = FU_NK;                        // layer-1 k-steps (80 taps in 5 x 16 slots; the 81st is an fp32 FMA)
constexpr float F_INV = 1.f / 256.f;             // undoes FusedF16Weights' 2^8 weight scale (exact)
#ifndef FU_WEAVE
#define FU_WEAVE 1
#endif
#ifndef FU_SEQ
#define FU_SEQ 1
#endif
#ifndef FU_PRIO
#define FU_PRIO 0      // wave priority in the layer-1 phase
#endif
#ifndef FU_PRIO23
#define FU_PRIO23 2    // ... and in the layers-2+3 phase
#endif

constexpr int L_W1 = 0;                                          // byte offsets into dynamic LDS
constexpr int L_W2 = L_W1 + (int)sizeof(FusedF16Weights::w1);
constexpr int L_W3 = L_W2 + (int)sizeof(FusedF16Weights::w2);
constexpr int L_B1 = L_W3 + (int)sizeof(FusedF16Weights::w3);
constexpr int L_B2 = L_B1 + 64 * 4;
constexpr int L_W88 = L_B2 + 32 * 4;
constexpr int L_Y = L_W88 + 64 * 4;
constexpr int L_P = L_Y + NW * RING;
constexpr int L_END = L_P + NW * PW * 4;
static_assert(L_Y % 16 == 0 && L_P % 16 == 0, "alignment");
static_assert(L_END <= 160 * 1024, "LDS budget");
static_assert(offsetof(FusedF16Weights, b1) == L_B1 && offsetof(FusedF16Weights, b2) == L_B2 &&
              offsetof(FusedF16Weights, w88) == L_W88 && offsetof(FusedF16Weights, b3) == L_Y, "blob head == LDS image");

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
// the fp16 tails: 4 instructions per pair.  Only the v_fma_mix_f32 is inline asm (the compiler folds fma(h, -1, x) to a
// subtraction of the unpacked head, 7 instructions per pair); both conversions are ordinary code, so every register an
// MFMA reads is written by an instruction the hazard recogniser can see.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float* x, h8& xh, h8& xl)
{
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const h2 hp = {(_Float16)x[2 * p], (_Float16)x[2 * p + 1]};
        const unsigned hb = __builtin_bit_cast(unsigned, hp);
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hb), "v"(x[2 * p]));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hb), "v"(x[2 * p + 1]));
        xh[2 * p] = hp[0]; xh[2 * p + 1] = hp[1];
        xl[2 * p] = (_Float16)r0; xl[2 * p + 1] = (_Float16)r1;
    }
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
    const FusedF16Weights* __restrict__ blob, int chunk_rows, int tiles_x, unsigned long long* __restrict__ dbg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const _Float16* W1f = reinterpret_cast<const _Float16*>(lds_raw + L_W1);
    const _Float16* W2f = reinterpret_cast<const _Float16*>(lds_raw + L_W2);
    const _Float16* W3f = reinterpret_cast<const _Float16*>(lds_raw + L_W3);
    const float* B1s = reinterpret_cast<const float*>(lds_raw + L_B1);     // biases stay in LDS: 48 VGPRs are worth more
    const float* B2s = reinterpret_cast<const float*>(lds_raw + L_B2);
    const float* W88s = reinterpret_cast<const float*>(lds_raw + L_W88);
    float* Pall = reinterpret_cast<float*>(lds_raw + L_P);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5, col = lane & 31;
    const int txi = blockIdx.x % tiles_x, cyi = blockIdx.x / tiles_x;
    const int GX0 = txi * GW;                                   // first output column of the workgroup
    const int R0 = out_row0 + cyi * chunk_rows;                 // output rows [R0, R1) of this chunk
    const int R1 = min(R0 + chunk_rows, out_row0 + out_rows);
    if (R0 >= R1) return;                                       // uniform per workgroup, before the barrier

    {   // weight fragments + biases: the head of the blob is the LDS image
        const uint4* src = reinterpret_cast<const uint4*>(blob);
        uint4* dst = reinterpret_cast<uint4*>(lds_raw);
        for (int e = tid; e < L_Y / 16; e += NT) dst[e] = src[e];
    }
    const float b3 = blob->b3;
    __syncthreads();                                            // the only workgroup barrier of the kernel

    const int cx0 = GX0 + OW * wv - 2;                          // image column of this wave's layer-2 column 0
    const int TX0 = cx0 - 4;                                    // image column of this wave's staged column 0
    const int A0 = clampi(R0 - 2, 0, H - 1);                    // first layer-2 row this chunk computes
    const int ubase = A0 - 4;                                   // virtual Y row held by ring slot 0 (mod 12)
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
                // Slot layout [hi0 | lo0 | hi1 | lo1]; copy 1 holds column t at t - 1.  A B fragment starts at column q0 of copy
                // q0 & 1, so even lanes 2k read dword k of copy 0 and odd lanes 2k+1 dword 2*PB/4 + k = 80 + k of copy 1: banks
                // [0,16) and [16,32) of the 32 a ds_read_b32 lane group sees -- disjoint.  (Round 2 had copy 1 at +PB with a +1
                // shift: dword 41 + k, banks 9..24, a 2-way conflict on every layer-1 fragment read: 20 % of all LDS cycles.)
                row[t] = hi;                      row[PB / 2 + t] = lo;
                if (t >= 1) { row[PB + t - 1] = hi; row[3 * PB / 2 + t - 1] = lo; }
            }
        }
    };
    auto wave_sync = [] {               // LDS hand-over between the lanes of ONE wave: no s_barrier involved
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // defensive: no fragment reads the ring's padding columns, but if one ever did (against zero weights) it must see
    // finite numbers, not whatever the previous kernel left in LDS
    for (int i = lane; i < RING / 4; i += 64) reinterpret_cast<unsigned*>(Yr)[i] = 0u;
    wave_sync();
    for (int k = 0; k < SLOTS / STAGE; ++k) { fetch(ubase + STAGE * k); land(ubase + STAGE * k); }
    wave_sync();
    // ---- per-lane constants ----
    float* Pw = Pall + wv * PW;
    int pidx[5];                                                // P column read for dx = 0..4 (clamped to the image)
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) pidx[dx] = clampi(clampi(cx0 + lane + dx - 2, 0, W - 1) - cx0, 0, 32 * SEG - 1);
    const int ox = cx0 + lane;                                  // output column of this lane
    const bool ox_ok = lane >= 2 && lane < 2 + OW && ox < W;
    // LDS byte address (inside a ring slot) of this lane's B fragment for segment 0: staged column q0, from the copy
    // whose alignment suits its parity; segment 1 is +64 B, the lo plane +PB.  Kept opaque so that every read below
    // is "one base register + immediate offset".
    const int q0 = col;
    unsigned frag_hi = (unsigned)(L_Y + wv * RING + ((q0 & 1) ? 2 * PB + 2 * (q0 - 1) : 2 * q0));
    unsigned frag_el = (unsigned)(L_Y + wv * RING + 2 * q0);     // element-wise reads: unshifted copy, any alignment
    asm volatile("" : "+v"(frag_hi), "+v"(frag_el));

    float O[5] = {0.f, 0.f, 0.f, 0.f, 0.f};                     // partial sums of output rows v+2 .. v-2
    const int nv = R1 - R0 + 4;                                 // virtual layer-2 rows R0-2 .. R1+1
    int st_cur = 0;                                             // ring stage: rows A0+4*st_cur .. +3 are computable
    if (A0 + STAGE <= min(R1 + 1, H - 1)) fetch(ubase + SLOTS);

    // ---------------------------------------------------------------------------------------------------------------
    // One pipeline step = layer 1 of row `a_nxt` (108 MFMAs, nothing else to do) woven together with layers 2+3, the P
    // hand-over and the gather of the PREVIOUS row (36 MFMAs and all of the VALU work): the previous row's work is cut
    // into nine slices, one per layer-1 k-step, so every 12-MFMA k-step has ~64 VALU instructions to issue in its
    // shadow and the matrix pipe never waits for a conversion chain.  Accumulators ping-pong between two sets.
    // ---------------------------------------------------------------------------------------------------------------
    f32x16 accA[SEG][2], accB[SEG][2];
    auto step = [&](auto DO_L1, auto DO_L23, f32x16 (&cur)[SEG][2], f32x16 (&nxt)[SEG][2], int a_nxt, int v_prev, int row_idx) {
        constexpr bool L1 = decltype(DO_L1)::value, L23 = decltype(DO_L23)::value;
        if constexpr (DIAG) {
            if (L1 && blockIdx.x == 0 && lane == 0 && row_idx < 64 && wv < 8 && dbg) dbg[((size_t)wv * 64 + row_idx) * 4] = __builtin_amdgcn_s_memtime();
        }
        int s0 = 0;
        h8 bh[SEG], bl[SEG], a0h, a0l, a1h, a1l;
        auto wrap = [&](int r) { return s0 + r >= SLOTS ? s0 + r - SLOTS : s0 + r; };       // ring slot of window row r
        auto load_step = [&](int s, h8 (&xbh)[SEG], h8 (&xbl)[SEG], h8& x0h, h8& x0l, h8& x1h, h8& x1l) {
            if (s < 4) {
                // window rows 2s (lanes 0-31) and 2s+1 (lanes 32-63), taps dx = 0..7: 8 consecutive halves of one row
                const unsigned off = (unsigned)((half ? wrap(2 * s + 1) : wrap(2 * s)) * SB);
                const unsigned* yh = reinterpret_cast<const unsigned*>(lds_raw + (frag_hi + off));
                const unsigned* yl = yh + PB / 4;
#pragma unroll
                for (int g = 0; g < SEG; ++g) {
                    u32x4 hi4, lo4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { hi4[q] = yh[16 * g + q]; lo4[q] = yl[16 * g + q]; }
                    xbh[g] = as_h8(hi4); xbl[g] = as_h8(lo4);
                }
            } else {
                // lanes 0-31: window row 8, taps dx = 0..7 (row-wise); lanes 32-63: tap column dx = 8 of window rows
                // 0..7 (column-wise).  One address per element and lane, 16-bit reads from the unshifted planes.
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned adr = frag_el + (unsigned)(half ? wrap(j) * SB + 16 : wrap(8) * SB + 2 * j);
                    const _Float16* e = reinterpret_cast<const _Float16*>(lds_raw + adr);
#pragma unroll
                    for (int g = 0; g < SEG; ++g) { xbh[g][j] = e[32 * g]; xbl[g][j] = e[32 * g + PB / 2]; }
                }
            }
            x0h = *reinterpret_cast<const h8*>(W1f + (((s * 2 + 0) * 2 + 0) * 64 + lane) * 8);
            x0l = *reinterpret_cast<const h8*>(W1f + (((s * 2 + 0) * 2 + 1) * 64 + lane) * 8);
            x1h = *reinterpret_cast<const h8*>(W1f + (((s * 2 + 1) * 2 + 0) * 64 + lane) * 8);
            x1l = *reinterpret_cast<const h8*>(W1f + (((s * 2 + 1) * 2 + 1) * 64 + lane) * 8);
        };
        // The previous row's work comes in "units" of 8 activations x 2 pieces -> 3 MFMAs; slice s of the step runs
        // the two units U[s][0..1].  kind 2 = layer 2 (W2 fragment blk*2+ks), kind 3 = layer 3 (W3 fragment ks).
        struct Unit { int kind, g, blk, ks; };
        constexpr Unit U[9][2] = {{{2, 0, 0, 0}, {2, 0, 0, 1}}, {{2, 0, 1, 0}, {2, 0, 1, 1}}, {{2, 1, 0, 0}, {2, 1, 0, 1}},
                                  {{2, 1, 1, 0}, {2, 1, 1, 1}}, {{3, 0, 0, 0}, {3, 0, 0, 1}}, {{3, 1, 0, 0}, {3, 1, 0, 1}},
                                  {{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
        // every LDS operand is requested one region before it is used (the compiler, left alone, re-loads operands right
        // before their use and the wave then sits out the LDS latency ~70 times per row)
        auto load_units = [&](int s, h8 (&f)[2][2]) {
            if constexpr (L23) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const Unit q = U[s][u];
                    if (q.kind == 2) {
                        f[u][0] = *reinterpret_cast<const h8*>(W2f + (((q.blk * 2 + q.ks) * 2 + 0) * 64 + lane) * 8);
                        f[u][1] = *reinterpret_cast<const h8*>(W2f + (((q.blk * 2 + q.ks) * 2 + 1) * 64 + lane) * 8);
                    } else if (q.kind == 3) {
                        f[u][0] = *reinterpret_cast<const h8*>(W3f + ((q.ks * 2 + 0) * 64 + lane) * 8);
                        f[u][1] = *reinterpret_cast<const h8*>(W3f + ((q.ks * 2 + 1) * 64 + lane) * 8);
                    }
                }
            }
        };
        f32x16 c1b[SEG][2] = {}, c2b = {};                         // the C operands that open the chains (biases x 2^8 ...)
        h8 fa[2][2] = {};
        if constexpr (L1) {
            if (((a_nxt - A0) >> 2) > st_cur) {                 // the row opens the next ring stage (uniform branch)
                wave_sync();                                    // this wave no longer reads the slots it overwrites
                land(ubase + SLOTS + STAGE * st_cur);
                ++st_cur;
                if (A0 + STAGE * (st_cur + 1) <= min(R1 + 1, H - 1)) fetch(ubase + SLOTS + STAGE * st_cur);
                wave_sync();
            }
            s0 = mod12(a_nxt - 4 - ubase);                      // ring slot of the first window row (uniform)
            load_step(0, bh, bl, a0h, a0l, a1h, a1l);
            // chain openers: bias + the 81st tap (8,8) as an exact fp32 FMA -- y88 = this lane's pixel of window row 8 at
            // dx = 8, rebuilt from its two fp16 pieces; 64 FMAs in a phase whose VALU slots are otherwise idle
            const _Float16* e88 = reinterpret_cast<const _Float16*>(lds_raw + (frag_el + (unsigned)(wrap(8) * SB) + 16));
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const f32x16 bias = *reinterpret_cast<const f32x16*>(B1s + half * 32 + 16 * blk);
                const f32x16 w88 = *reinterpret_cast<const f32x16*>(W88s + half * 32 + 16 * blk);
#pragma unroll
                for (int g = 0; g < SEG; ++g) {
                    const float y88 = (float)e88[32 * g] + (float)e88[32 * g + PB / 2];
#pragma unroll
                    for (int r = 0; r < 16; ++r) c1b[g][blk][r] = __builtin_fmaf(w88[r], y88, bias[r]);
                }
            }
        }
        if constexpr (L23) c2b = *reinterpret_cast<const f32x16*>(B2s + half * 16);
        load_units(0, fa);
        __builtin_amdgcn_sched_barrier(0);

        f32x16 acc2[SEG] = {}, accp[SEG] = {};
        auto run_unit = [&](Unit q, const h8 (&f)[2]) {
            float x[8];
            if (q.kind == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = fmaxf(cur[q.g][q.blk][8 * q.ks + j] * F_INV, 0.f);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = fmaxf(acc2[q.g][8 * q.ks + j] * F_INV, 0.f);
            }
            h8 xh, xl;
            split8(x, xh, xl);
            f32x16& d = q.kind == 2 ? acc2[q.g] : accp[q.g];
            const bool opens = q.kind == 2 && q.blk == 0 && q.ks == 0;          // layer-2 chain starts from its bias
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], xh, opens ? c2b : d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0], xl, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1], xh, d, 0, 0, 0);
        };
        auto pw = [&](int g) {                                  // P rows of a segment: register r of half h is tap 8*(r/4)+4h+r%4
#pragma unroll
            for (int r = 0; r < 13; ++r) {
                const int t0 = 8 * (r >> 2) + (r & 3);
                if (r < 12 || half == 0) Pw[(t0 + 4 * half) * PS + 32 * g + col] = accp[g][r];       // x 2^8; undone once per output pixel
            }
        };
        auto slice = [&](int s) {
            if constexpr (L23) {
                if (s == 5) pw(0);
                if (s == 6) pw(1);
                if (U[s][0].kind) { run_unit(U[s][0], fa[0]); run_unit(U[s][1], fa[1]); }
                if (s == 7) {
                    wave_sync();
                    // layer-3 gather: layer-2 row v feeds output rows v+2-dy with tap row dy
#pragma unroll
                    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 5; ++dx) O[dy] += Pw[(dy * 5 + dx) * PS + pidx[dx]];
                }
                if (s == 8) {
                    const int orow = v_prev - 2;                // complete: its last contribution was tap row 4
                    if (orow >= R0 && ox_ok) out[(size_t)(orow - out_row0) * W + ox] = fminf(fmaxf(__builtin_fmaf(O[4], F_INV, b3), 0.f), 255.f);
                    O[4] = O[3]; O[3] = O[2]; O[2] = O[1]; O[1] = O[0]; O[0] = 0.f;
                    __builtin_amdgcn_wave_barrier();            // the next row overwrites Pw
                }
            }
        };
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            h8 nbh[SEG], nbl[SEG], n0h, n0l, n1h, n1l, fn[2][2] = {};
            if constexpr (L1) { if (s + 1 < NK) load_step(s + 1, nbh, nbl, n0h, n0l, n1h, n1l); }
            if (s + 1 < 9) load_units(s + 1, fn);
            __builtin_amdgcn_sched_barrier(0);                  // requests first: they may not sink towards their uses
            slice(s);
            if (L1 && s < NK) {
#pragma unroll
                for (int g = 0; g < SEG; ++g) {
                    nxt[g][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bh[g], s == 0 ? c1b[g][0] : nxt[g][0], 0, 0, 0);
                    nxt[g][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bh[g], s == 0 ? c1b[g][1] : nxt[g][1], 0, 0, 0);
                    nxt[g][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bl[g], nxt[g][0], 0, 0, 0);
                    nxt[g][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bl[g], nxt[g][1], 0, 0, 0);
                    nxt[g][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, bh[g], nxt[g][0], 0, 0, 0);
                    nxt[g][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, bh[g], nxt[g][1], 0, 0, 0);
                }
            }
            if (L1 && L23 && FU_WEAVE && s < 7) {
                // weave the region: one MFMA, then up to five VALU in its shadow (an MFMA holds the issue port for 8 of
                // its 32 cycles; ~5 other instructions fit in the rest)
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                  // slices stay with their k-step
            if (s + 1 < 9) {
                if (L1 && s + 1 < NK) {
#pragma unroll
                    for (int g = 0; g < SEG; ++g) { bh[g] = nbh[g]; bl[g] = nbl[g]; }
                    a0h = n0h; a0l = n0l; a1h = n1h; a1l = n1l;
                }
                fa[0][0] = fn[0][0]; fa[0][1] = fn[0][1]; fa[1][0] = fn[1][0]; fa[1][1] = fn[1][1];
            }
        }
    };
    // Layers 2+3, the P hand-over and the gather of one row as ONE scheduling region (sequential form): all W2/W3
    // fragments are requested up front (both segments share them: 12 ds_read_b128), and the two segments' accumulator
    // chains are interleaved unit by unit, so one chain's conversions run in the shadow of the other chain's MFMAs.
    auto l23 = [&](f32x16 (&cur)[SEG][2], int v_prev) {
        h8 w2h[2][2], w2l[2][2], w3h[2], w3l[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                w2h[blk][ks] = *reinterpret_cast<const h8*>(W2f + (((blk * 2 + ks) * 2 + 0) * 64 + lane) * 8);
                w2l[blk][ks] = *reinterpret_cast<const h8*>(W2f + (((blk * 2 + ks) * 2 + 1) * 64 + lane) * 8);
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            w3h[ks] = *reinterpret_cast<const h8*>(W3f + ((ks * 2 + 0) * 64 + lane) * 8);
            w3l[ks] = *reinterpret_cast<const h8*>(W3f + ((ks * 2 + 1) * 64 + lane) * 8);
        }
        const f32x16 c2b = *reinterpret_cast<const f32x16*>(B2s + half * 16);
        f32x16 acc2[SEG], accp[SEG] = {};
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int g = 0; g < SEG; ++g) {
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[j] = fmaxf(cur[g][blk][8 * ks + j] * F_INV, 0.f);
                    h8 xh, xl;
                    split8(x, xh, xl);
                    acc2[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[blk][ks], xh, (blk | ks) ? acc2[g] : c2b, 0, 0, 0);
                    acc2[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2h[blk][ks], xl, acc2[g], 0, 0, 0);
                    acc2[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2l[blk][ks], xh, acc2[g], 0, 0, 0);
                }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int g = 0; g < SEG; ++g) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = fmaxf(acc2[g][8 * ks + j] * F_INV, 0.f);
                h8 xh, xl;
                split8(x, xh, xl);
                accp[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3h[ks], xh, accp[g], 0, 0, 0);
                accp[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3h[ks], xl, accp[g], 0, 0, 0);
                accp[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3l[ks], xh, accp[g], 0, 0, 0);
            }
        // P rows: register r of half h is tap 8*(r/4) + 4h + r%4; taps >= 25 are padding
#pragma unroll
        for (int g = 0; g < SEG; ++g)
#pragma unroll
            for (int r = 0; r < 13; ++r) {
                const int t0 = 8 * (r >> 2) + (r & 3);
                if (r < 12 || half == 0) Pw[(t0 + 4 * half) * PS + 32 * g + col] = accp[g][r];       // x 2^8; undone once per output pixel
            }
        wave_sync();
        // layer-3 gather: layer-2 row v feeds output rows v+2-dy with tap row dy
#pragma unroll
        for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int dx = 0; dx < 5; ++dx) O[dy] += Pw[(dy * 5 + dx) * PS + pidx[dx]];
        const int orow = v_prev - 2;                            // complete: its last contribution was tap row 4
        if (orow >= R0 && ox_ok) out[(size_t)(orow - out_row0) * W + ox] = fminf(fmaxf(__builtin_fmaf(O[4], F_INV, b3), 0.f), 255.f);
        O[4] = O[3]; O[3] = O[2]; O[2] = O[1]; O[1] = O[0]; O[0] = 0.f;
        __builtin_amdgcn_wave_barrier();                        // the next row overwrites Pw
    };

    using T = std::true_type;
    using F = std::false_type;
    auto row_of = [&](int k) { return clampi(R0 - 2 + k, 0, H - 1); };    // the reference clamps layer-2 ACTIVATIONS at the border

#if FU_SEQ || FU_SEG != 2
    static_assert(FU_SEQ, "the cross-row pipelined form (FU_SEQ=0) is written for two segments per wave");
    // sequential form (A/B experiment): layer 1 of a row, then its layers 2+3 -- no cross-row overlap, half the accumulators
#pragma unroll 1
    for (int k = 0; k < nv; ++k) {
        // Phase-dependent priority.  Without it the older wave of a SIMD pair wins every arbitration (9.5k vs 16.7k
        // cycles per row; the workgroup waits for the starved half).  The wave in the layers-2+3 phase -- few MFMAs, each
        // at the end of a conversion chain -- outranks its partner; the partner's MFMA-only layer 1 fills every other
        // slot of the matrix pipe.  (The opposite assignment also balances the pair but is 2-5 % slower: the conversion
        // chains then wait behind 72 back-to-back MFMAs.)
        __builtin_amdgcn_s_setprio(FU_PRIO);
        step(T{}, F{}, accB, accA, row_of(k), 0, k);
        __builtin_amdgcn_s_setprio(FU_PRIO23);
        if constexpr (DIAG) {           // phase boundary stamp (after the last layer-1 MFMA has been issued)
            if (blockIdx.x == 0 && lane == 0 && k < 64 && wv < 8 && dbg) dbg[((size_t)wv * 64 + k) * 4 + 1] = __builtin_amdgcn_s_memtime();
        }
        l23(accA, R0 - 2 + k);
    }
#else
    step(T{}, F{}, accB, accA, row_of(0), 0, 0);                           // prologue: layer 1 of the first row -> accA
    int k = 1;
#pragma unroll 1
    for (; k + 1 < nv; k += 2) {
        step(T{}, T{}, accA, accB, row_of(k), R0 - 2 + k - 1, k);
        step(T{}, T{}, accB, accA, row_of(k + 1), R0 - 2 + k, k + 1);
    }
    if (k < nv) {
        step(T{}, T{}, accA, accB, row_of(k), R0 - 2 + k - 1, k);
        step(F{}, T{}, accB, accA, 0, R0 - 2 + k, k + 1);                  // epilogue
    } else {
        step(F{}, T{}, accA, accB, 0, R0 - 2 + k - 1, k);
    }
#endif
}

hipError_t fused_f16_prepare()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_f16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, L_END);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_f16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, L_END);
}

void launch_fused_f16(const float* Y, int W, int H, int y_row_base, int y_rows, float* out, int out_row0, int out_rows,
                      const FusedF16Weights* d_blob, int num_cus, hipStream_t s, unsigned long long* dbg)
{
    if (out_rows <= 0) return;
    const int tiles_x = (W + GW - 1) / GW;
    // one workgroup per CU: cut the rows into just enough chunks to fill the chip (a chunk start costs 4 warm-up rows)
    int chunks = std::max(1, num_cus / tiles_x);          // never more workgroups than CUs: one round, no tail
    chunks = std::min(chunks, std::max(1, out_rows / 16));
    const int chunk_rows = (out_rows + chunks - 1) / chunks;
    chunks = (out_rows + chunk_rows - 1) / chunk_rows;
    if (dbg)
        hipLaunchKernelGGL(k_fused_f16<true>, dim3(tiles_x * chunks), dim3(NT), L_END, s, Y, W, H, y_row_base, y_rows, out, out_row0,
                           out_rows, d_blob, chunk_rows, tiles_x, dbg);
    else
        hipLaunchKernelGGL(k_fused_f16<false>, dim3(tiles_x * chunks), dim3(NT), L_END, s, Y, W, H, y_row_base, y_rows, out, out_row0,
                           out_rows, d_blob, chunk_rows, tiles_x, dbg);
}

}  // namespace srcnn
