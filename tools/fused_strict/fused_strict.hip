// fused_strict.hip -- EXPERIMENT ("measure the strict fusion instead of arguing it").  Not part of the product.
//
// k_fused_strict: layers 1+2+3 of the SRCNN Y path in ONE strict (bit-exact) kernel -- the 32 layer-2 planes never go to HBM.
// Reference arithmetic reproduced exactly as the two production kernels do (src/libsrcnn.cpp:350-447 for layers 1+2: rounded
// product from the K=1 MFMA with C = 0, rounded add on the VALU, tap / channel order kept; :449-529 for layer 3: fp32 product,
// fp64 sum per channel in window order, fp32 running sum over the channels, bias, clamp).
//
// Geometry (what 160 KB of LDS allow):
//   * a workgroup (8 waves) owns a strip of 124 output columns and a chunk of rows and MARCHES down it two rows per step;
//   * phase A of a step: the 8 waves compute the 8 segments (32 px) of layer-2 rows q, q+1 over 128 columns (124 + 2 halo
//     columns each side, recomputed by the neighbouring strip: +3.2 %) into a ring of SIX fp32 layer-2 rows in LDS
//     (32 planes x 130 floats per row: 99.8 KB);
//   * phase B: output rows q-2, q-1 (they need layer-2 rows q-4 .. q+1 = the whole ring), 256 pixels: waves 0-3 take channels
//     0-15 of 64 pixels each, waves 4-7 channels 16-31 of the same pixels -- the channel is wave-uniform, so the weights are
//     scalar operands (a first version split the channels over the lane halves and read weights AND window from LDS: the LDS
//     pipe, not the VALU, set its pace: 12.5 ms per frame).  The fp32 running sum over the channels is sequential: waves 0-3
//     hand their sum after channel 15 to waves 4-7 through 1 KB of LDS behind a third barrier;
//   * clamp-to-edge of the ACTIVATIONS at the true image border is done by addressing (clamped ring row / column), never by
//     computing layers 1+2 outside the image;
//   * the per-wave transposition slab of layer 2 holds 16 channels at a time (2 KB instead of the production kernel's 4).
// LDS: layer-1+2 weights 29.3 KB + ring 99.8 KB + slabs 16 KB + Y tile 5.4 KB + hand-over 1 KB = 151.5 KB -> one workgroup per CU,
// 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#pragma clang fp contract(off)

typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PIN(v) asm volatile("" : "+v"(v))

namespace {
constexpr int NW = 8, NT = 64 * NW;
constexpr int SW = 128, OW = 124;                 // computed layer-2 columns / output columns of a strip
constexpr int PS = 130, NR = 6;                   // ring: plane stride (floats), row slots
constexpr int YW = SW + 8, YR = 2 + 8;            // Y tile of one step
constexpr int N_W1 = 81 * 64, N_W2 = 32 * 64, N_B1 = 64, N_B2 = 32, N_W3 = 32 * 30;
constexpr int N_WLDS = N_W1 + N_W2 + N_B1 + N_B2;                 // what is staged in LDS
constexpr int N_WIMG = N_WLDS + N_W3;                             // the global image: + layer 3 (scalar loads) + b3 behind it
constexpr int N_RING = NR * 32 * PS, N_SLAB = NW * 512, N_YT = YR * YW, N_S15 = 2 * SW;
constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(N_WLDS + N_RING + N_SLAB + N_YT + N_S15);

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(NT, 2) void k_fused_strict(const float* __restrict__ Y, int W, int H, float* __restrict__ out,
                                                        const float* __restrict__ wimg, int rows_per_chunk, int phases)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* W1s = lds;                       // [tap][lane = channel]
    float* W2s = W1s + N_W1;                // [f/2][lane]: lanes 0-31 w2[m = lane][f], lanes 32-63 w2[m = lane - 32][f + 1]
    float* B1s = W2s + N_W2;                // [half][reg]
    float* B2s = B1s + N_B1;                // [half][reg]
    float* ring = B2s + N_B2;               // [slot][m][PS]
    float* slab = ring + N_RING;            // [wave][16 channels][32 px]
    float* ytile = slab + N_SLAB;           // [YR][YW]
    float* s15 = ytile + N_YT;              // [2 rows][SW]: the running sum after channel 15

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = lane >> 5, col = lane & 31;
    for (int e = tid; e < N_WLDS; e += NT) lds[e] = wimg[e];
    const float b3 = wimg[N_WIMG];

    const int X0 = blockIdx.x * OW;
    const int R0 = blockIdx.y * rows_per_chunk, R1 = min(R0 + rows_per_chunk, H);
    if (R0 >= H) return;
    float* mySlab = slab + wv * 512;
    const f32x32 zero32 = {};

    // phase A roles: layer-2 row (q + arow), segment aseg of the strip
    const int arow = wv >> 2, aseg = wv & 3;
    // phase B roles: output row (q - 2 + brow), pixel k of the strip, channel group grp (wave-uniform)
    const int grp = __builtin_amdgcn_readfirstlane(wv >> 2);
    const int brow = (wv & 3) >> 1, k = (wv & 1) * 64 + lane;
    const int x = X0 + k;
    const bool live = k < OW && x < W;
    int jj[5];
#pragma unroll
    for (int dx = 0; dx < 5; ++dx) jj[dx] = live ? clampi(x + dx - 2, 0, W - 1) - (X0 - 2) : 0;

    auto ytile_src = [&](int q, int e) {
        const int r = e / YW, c = e - r * YW;
        return Y + (size_t)clampi(q - 4 + r, 0, H - 1) * W + clampi(X0 - 6 + c, 0, W - 1);
    };
    int q = R0 - 2;
    for (int e = tid; e < N_YT; e += NT) ytile[e] = *ytile_src(q, e);
    __syncthreads();

    for (; q - 2 < R1; q += 2) {
        // ---------------- phase A: layers 1+2 for layer-2 row yc, 32 px ----------------
        const int yc = q + arow;
        if ((phases & 1) && yc >= 0 && yc < H && yc <= R1 + 1) {
            const float* yrow = ytile + arow * YW + aseg * 32 + col;
            f32x32 acc = zero32;
            {
                float a1 = W1s[lane], b1 = yrow[0];
                float a2 = W1s[64 + lane], b2 = yrow[1];
#pragma unroll
                for (int t = 0; t < 81; ++t) {
                    f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a1, b1, zero32, 0, 0, 0);
                    PIN(d);
                    a1 = a2; b1 = b2;
                    if (t + 2 < 81) {
                        a2 = W1s[(t + 2) * 64 + lane];
                        b2 = yrow[((t + 2) / 9) * YW + ((t + 2) % 9)];
                    }
                    acc += d;
                    PIN(acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            f32x16 acc2 = {};
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {                       // 16 channels at a time: 0-15, 16-31, 32-47, 48-63
                const int blk = qt >> 1, hf = qt & 1;
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = 8 * hf + r8;
                    const float v = fmaxf(acc[16 * blk + r] + B1s[half * 32 + 16 * blk + r], 0.f);
                    const int fl = 8 * (r8 >> 2) + (r8 & 3) + 4 * half;        // channel within this group of 16
                    mySlab[fl * 32 + col] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const float* w2p = W2s + qt * 8 * 64;
                float a1 = w2p[lane], b1 = mySlab[lane];
                float a2 = w2p[64 + lane], b2 = mySlab[64 + lane];
#pragma unroll
                for (int fp = 0; fp < 8; ++fp) {
                    f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a1, b1, zero32, 0, 0, 0);
                    PIN(d);
                    a1 = a2; b1 = b2;
                    if (fp + 2 < 8) { a2 = w2p[(fp + 2) * 64 + lane]; b2 = mySlab[(fp + 2) * 64 + lane]; }
                    acc2 += __builtin_shufflevector(d, d, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
                    acc2 += __builtin_shufflevector(d, d, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31);
                    PIN(acc2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_wave_barrier();
            }
            float* rrow = ring + (size_t)((yc % NR) * 32) * PS + aseg * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 8 * (r >> 2) + (r & 3) + 4 * half;
                rrow[m * PS] = fmaxf(acc2[r] + B2s[half * 16 + r], 0.f);
            }
        }
        __syncthreads();                                           // ring rows q, q+1 complete; the Y tile is free
        // the next step's Y tile is requested now and lands after phase B
        float yp[(N_YT + NT - 1) / NT];
        const bool more = q + 2 - 2 < R1;
#pragma unroll
        for (int i = 0; i < (N_YT + NT - 1) / NT; ++i) {
            const int e = tid + i * NT;
            yp[i] = (more && e < N_YT) ? *ytile_src(q + 2, e) : 0.f;
        }
        // ---------------- phase B: layer 3 for output row orow ----------------
        const int orow = q - 2 + brow;
        const bool act = (phases & 2) && orow >= R0 && orow < R1;   // wave-uniform
        double a[16];
        if (act) {
            int sb[5];
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) sb[dy] = (clampi(orow + dy - 2, 0, H - 1) % NR) * 32 * PS;
            const float* rb = ring + grp * 16 * PS;
            const float* wg = wimg + N_WLDS + grp * 16 * 30;       // wave-uniform: scalar loads
#pragma unroll
            for (int ml = 0; ml < 16; ++ml) {
                const float* pl = rb + ml * PS;
                const float* wr = wg + ml * 30;
                double s = 0.0;
#pragma unroll
                for (int dy = 0; dy < 5; ++dy) {
                    const float* row = pl + sb[dy];
#pragma unroll
                    for (int dx = 0; dx < 5; ++dx) s = s + (double)(wr[dy * 6 + dx] * row[jj[dx]]);
                }
                a[ml] = s;
            }
            if (grp == 0) {
                float sum = 0.f;
#pragma unroll
                for (int ml = 0; ml < 16; ++ml) sum = (float)((double)sum + a[ml]);
                s15[brow * SW + k] = sum;
            }
        }
        __syncthreads();                                           // the sums after channel 15 are in place
        if (act && grp == 1) {
            float sum = s15[brow * SW + k];
#pragma unroll
            for (int ml = 0; ml < 16; ++ml) sum = (float)((double)sum + a[ml]);
            if (live) {
                float v = sum + b3;
                v = fminf(fmaxf(v, 0.f), 255.f);
                out[(size_t)orow * W + x] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < (N_YT + NT - 1) / NT; ++i) {
            const int e = tid + i * NT;
            if (e < N_YT) ytile[e] = yp[i];
        }
        __syncthreads();                                           // phase B is done with the ring; the next Y tile is in place
    }
}

float* g_wimg = nullptr;
}  // namespace

extern "C" {

// weights: the 8129 floats of tests/golden/weights_f32.bin (b1, W1[k][i][j], b2, W2[m][f], b3, W3[m][x][y]) -- src/convdata.h
int fused_strict_init(const float* w)
{
    std::vector<float> img(N_WIMG + 1, 0.f);
    const float* b1 = w; const float* w1 = b1 + 64; const float* b2 = w1 + 64 * 81; const float* w2 = b2 + 32;
    const float* b3 = w2 + 32 * 64; const float* w3 = b3 + 1;
    float* W1s = img.data(); float* W2s = W1s + N_W1; float* B1s = W2s + N_W2; float* B2s = B1s + N_B1; float* W3s = B2s + N_B2;
    for (int t = 0; t < 81; ++t) for (int kch = 0; kch < 64; ++kch) W1s[t * 64 + kch] = w1[kch * 81 + t];
    for (int fp = 0; fp < 32; ++fp) for (int l = 0; l < 64; ++l) W2s[fp * 64 + l] = w2[(l & 31) * 64 + 2 * fp + (l >> 5)];
    for (int hf = 0; hf < 2; ++hf) for (int r = 0; r < 32; ++r) B1s[hf * 32 + r] = b1[32 * (r >> 4) + 8 * ((r & 15) >> 2) + 4 * hf + (r & 3)];
    for (int hf = 0; hf < 2; ++hf) for (int r = 0; r < 16; ++r) B2s[hf * 16 + r] = b2[8 * (r >> 2) + 4 * hf + (r & 3)];
    for (int m = 0; m < 32; ++m) for (int dy = 0; dy < 5; ++dy) for (int dx = 0; dx < 5; ++dx) W3s[m * 30 + dy * 6 + dx] = w3[m * 25 + dx * 5 + dy];
    img[N_WIMG] = *b3;
    if (!g_wimg && hipMalloc((void**)&g_wimg, img.size() * sizeof(float)) != hipSuccess) return -1;
    if (hipMemcpy(g_wimg, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return -2;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused_strict), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess) return -3;
    return 0;
}

// d_up: the upscaled Y plane (W x H); d_out: Y' (W x H).  chunks <= 0: as many row chunks as keep one round of workgroups.
int fused_strict_run(const float* d_up, int W, int H, float* d_out, int chunks, void* stream)
{
    if (!g_wimg) return -1;
    const int strips = (W + OW - 1) / OW;
    int ncu = 256;
    if (chunks <= 0) chunks = ncu / strips > 0 ? ncu / strips : 1;
    int rows = (H + chunks - 1) / chunks;
    rows = (rows + 1) & ~1;
    if (rows < 2) rows = 2;
    chunks = (H + rows - 1) / rows;
    static const int phases = [] { const char* e = getenv("FS_PHASES"); return e ? atoi(e) : 3; }();      // timing experiments: 1 = layers 1+2 only, 2 = layer 3 only
    hipLaunchKernelGGL(k_fused_strict, dim3(strips, chunks), dim3(NT), LDS_BYTES, (hipStream_t)stream, d_up, W, H, d_out, g_wimg, rows, phases);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int fused_strict_lds_bytes(void) { return (int)LDS_BYTES; }

}  // extern "C"
