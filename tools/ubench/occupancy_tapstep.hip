// occupancy_tapstep.hip -- round 6: does MORE OCCUPANCY buy the strict tap-step anything?
// A strict tap-step is  d = mfma(a, b, 0)  (rounded products)  followed by  acc += d  (rounded sums, VALU).
// k_conv12_mfma uses v_mfma_f32_32x32x1_2b (32 acc + 32 d registers -> 126 VGPRs -> 4 waves per SIMD) and pays
// ~72 cycles per 1024 MAC (ideal 64: the fp32 MFMA and the fp32 VALU do not overlap, profiles/r01_mfma_coissue.txt).
// v_mfma_f32_16x16x1_4b needs 16 + 16 registers: up to 8 waves per SIMD.  profiles/r05_valu_add_patterns.txt shows the adds alone
// falling from 78 to 68 cycles per 2048 between 4 and 6 waves.  This probe measures the whole step, single result buffer
// exactly as the production kernel (the wave's adds wait for its own MFMA, the SIMD's other waves fill the gap):
//   M32      mfma32x32x1_2b + 16 v_pk_add_f32                       waves/SIMD 2 3 4
//   M16      mfma16x16x1_4b +  8 v_pk_add_f32                       waves/SIMD 2 3 4 5 6 8
//   M16L     M16 + the two ds_read_b32 operand fetches (two steps ahead) and the s_waitcnt of the real kernel
//   M16S     mfma16x16x1_4b + 16 v_add_f32 (unpacked)
//   M16D     M16 with two result buffers (software-pipelined: adds of step t beside the MFMA of step t+1)
// Wall-time based (the dispatcher need not spread workgroups evenly): every SIMD executes iters * WPS steps in `ms`.
// Output: cycles per 1024 MAC per SIMD.  Go / no-go for a 16-pixel-segment k_conv12_mfma: <= 67 at 6 or 8 waves.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)
#define PIN(v) asm volatile("" : "+v"(v))
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { M32, M16, M16L, M16S, M16D, NARM };
const char* kName[NARM] = {"mfma32x32x1_2b + 16 pk_add", "mfma16x16x1_4b + 8 pk_add", "mfma16 + 8 pk_add + 2 ds_read", "mfma16 + 16 v_add_f32",
                           "mfma16 + 8 pk_add, 2 buffers"};
constexpr int kMacs[NARM] = {2048, 1024, 1024, 1024, 1024};

template <int ARM, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, unsigned long long* clk, int iters, float a0, float b0)
{
    const int lane = threadIdx.x & 63;
    float a = a0 + lane * 1e-3f, b = b0 + lane * 2e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float res = 0.f;
    if constexpr (ARM == M32) {
        const f32x32 z = {};
        f32x32 acc = z;
        for (int it = 0; it < iters; ++it) {
            f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, z, 0, 0, 0); PIN(d);
            acc += d; PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 32; ++i) res += acc[i];
    } else if constexpr (ARM == M16) {
        const f32x16 z = {};
        f32x16 acc = z;
        for (int it = 0; it < iters; ++it) {
            f32x16 d = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, z, 0, 0, 0); PIN(d);
            acc += d; PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 16; ++i) res += acc[i];
    } else if constexpr (ARM == M16L) {
        __shared__ float opnd[16 * 64 + 24 * 72];          // 11 KB: 8 blocks per CU still fit
        for (int i = threadIdx.x; i < 16 * 64 + 24 * 72; i += 256) opnd[i] = a0 + 1e-3f * (i & 127);
        __syncthreads();
        const f32x16 z = {};
        f32x16 acc = z;
        const float* wa = opnd + lane;
        const float* xb = opnd + 16 * 64 + (lane & 15);
        float a1 = wa[0], b1 = xb[0], a2 = wa[64], b2 = xb[1];
        for (int it = 0; it < iters; it += 81) {
#pragma unroll
            for (int t = 0; t < 81; ++t) {
                f32x16 d = __builtin_amdgcn_mfma_f32_16x16x1f32(a1, b1, z, 0, 0, 0); PIN(d);
                a1 = a2; b1 = b2;
                a2 = wa[((t + 2) % 16) * 64];
                b2 = xb[((t + 2) / 9) * 72 + (t + 2) % 9];
                acc += d; PIN(acc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int i = 0; i < 16; ++i) res += acc[i];
    } else if constexpr (ARM == M16S) {
        const f32x16 z = {};
        f32x16 acc = z;
        for (int it = 0; it < iters; ++it) {
            f32x16 d = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, z, 0, 0, 0); PIN(d);
#pragma unroll
            for (int i = 0; i < 16; ++i) { float x = acc[i]; asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(d[i])); acc[i] = x; }
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 16; ++i) res += acc[i];
    } else {
        const f32x16 z = {};
        f32x16 acc = z, d1 = z;
        for (int it = 0; it < iters; it += 2) {
            f32x16 d0 = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, z, 0, 0, 0); PIN(d0);
            __builtin_amdgcn_sched_barrier(0);
            acc += d1; PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x1f32(b, a, z, 0, 0, 0); PIN(d1);
            __builtin_amdgcn_sched_barrier(0);
            acc += d0; PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 16; ++i) res += acc[i] + d1[i];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int ARM, int WPS>
void run(int cus, float* d_out, unsigned long long* d_clk)
{
    const int iters = 81 * 600, grid = cus * WPS;
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k<ARM, WPS>, 256, 0);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k<ARM, WPS>));
    hipLaunchKernelGGL((k<ARM, WPS>), dim3(grid), dim3(256), 0, 0, d_out, d_clk, 162, 1.0f, 0.5f);
    hipDeviceSynchronize();
    float best = 1e30f;
    std::vector<unsigned long long> c(2 * grid);
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<ARM, WPS>), dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; hipMemcpy(c.data(), d_clk, grid * 16, hipMemcpyDeviceToHost); }
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    std::vector<double> mhz(grid);
    for (int i = 0; i < grid; ++i) mhz[i] = (double)c[2 * i] / (double)c[2 * i + 1] * 100.0;
    std::sort(mhz.begin(), mhz.end());
    const double ns_per_step = best * 1e6 / ((double)iters * WPS);
    const double cyc = ns_per_step * mhz[grid / 2] * 1e-3;
    printf("%-32s waves/SIMD=%d (resident blocks/CU %d, %3d VGPR)  %7.2f ms  clock %4.0f MHz  %6.1f cycles/step/SIMD = %5.1f cycles per 1024 MAC\n",
           kName[ARM], WPS, occ, fa.numRegs, best, mhz[grid / 2], cyc, cyc * 1024.0 / kMacs[ARM]);
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d_out; hipMalloc(&d_out, sizeof(float) * 256 * cus * 8);
    unsigned long long* d_clk; hipMalloc(&d_clk, 16 * cus * 8);
    printf("%s, %d CUs; ideal 64 cycles per 1024 strict MAC per SIMD (32 MFMA + 32 add)\n", p.gcnArchName, cus);
    run<M32, 2>(cus, d_out, d_clk); run<M32, 3>(cus, d_out, d_clk); run<M32, 4>(cus, d_out, d_clk); printf("\n");
    run<M16, 2>(cus, d_out, d_clk); run<M16, 3>(cus, d_out, d_clk); run<M16, 4>(cus, d_out, d_clk); run<M16, 5>(cus, d_out, d_clk);
    run<M16, 6>(cus, d_out, d_clk); run<M16, 8>(cus, d_out, d_clk); printf("\n");
    run<M16L, 4>(cus, d_out, d_clk); run<M16L, 6>(cus, d_out, d_clk); run<M16L, 8>(cus, d_out, d_clk); printf("\n");
    run<M16S, 4>(cus, d_out, d_clk); run<M16S, 6>(cus, d_out, d_clk); run<M16S, 8>(cus, d_out, d_clk); printf("\n");
    run<M16D, 4>(cus, d_out, d_clk); run<M16D, 6>(cus, d_out, d_clk); run<M16D, 8>(cus, d_out, d_clk); printf("\n");
    return 0;
}
