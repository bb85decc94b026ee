// f16_coissue.hip -- does the fp16 matrix pipe run beside VALU / LDS work on gfx950?  (The fp32 K=1 MFMA does not:
// profiles/r01_mfma_coissue.txt.)  This decides how the fused non-parity kernel schedules its split/convert
// VALU work: hidden under v_mfma_f32_32x32x16_f16 or paid on top of it.
// Variants per iteration: 4 MFMA (2 independent accumulator chains) | NV VALU ops | both interleaved in one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define PIN(v) asm volatile("" : "+v"(v))

enum { MFMA_ONLY, FMA_ONLY, CVT_ONLY, BOTH_FMA, BOTH_CVT, LDS_ONLY, BOTH_LDS, NV };
const char* kN[NV] = {"4 mfma_32x32x16_f16", "32 v_fma_f32", "32 v_cvt_f16_f32", "4 mfma + 32 v_fma_f32", "4 mfma + 32 v_cvt_f16_f32",
                      "8 ds_read_b128", "4 mfma + 8 ds_read_b128"};

template <int V>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters, float a0)
{
    __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 8];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 256 * 4 * 8; i += 256) lds[i] = i * 1e-6f;
    __syncthreads();
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(a0 + lane * 1e-3f + j); b[j] = (_Float16)(0.5f + j * 0.01f); }
    f32x16 acc0 = {}, acc1 = {};
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = lane * 1e-6f * i;
    float4 l[8];
    for (int i = 0; i < 8; ++i) l[i] = make_float4(0, 0, 0, 0);
    const float4* lp = reinterpret_cast<const float4*>(lds) + threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        constexpr bool M = (V == MFMA_ONLY || V == BOTH_FMA || V == BOTH_CVT || V == BOTH_LDS);
        constexpr bool F = (V == FMA_ONLY || V == BOTH_FMA), Cv = (V == CVT_ONLY || V == BOTH_CVT);
        constexpr bool Ld = (V == LDS_ONLY || V == BOTH_LDS);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (M) {
                if (q & 1) { acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0); }
                else { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0); }
            }
            if constexpr (F) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q * 8 + i]) : "v"(a0));
            }
            if constexpr (Cv) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(v[q * 8 + i]) : "v"(a0));
            }
            if constexpr (Ld) {
                l[2 * q] = lp[(2 * q) * 256]; l[2 * q + 1] = lp[(2 * q + 1) * 256];
            }
        }
        if constexpr (Ld) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(l[i].x), "+v"(l[i].y), "+v"(l[i].z), "+v"(l[i].w));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += l[i].x + l[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int V>
void run(int bpc, int cus, float* d_out, unsigned long long* d_clk)
{
    const int iters = 50000, grid = cus * bpc;
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, 500, 1.0f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(2 * grid);
    hipMemcpy(c.data(), d_clk, c.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> mhz(grid), cyc(grid);
    for (int i = 0; i < grid; ++i) { mhz[i] = (double)c[2 * i] / (double)c[2 * i + 1] * 100.0; cyc[i] = (double)c[2 * i] / iters; }
    std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
    printf("%-32s waves/SIMD=%d %8.2f ms  clock=%5.0f MHz  cycles/iter/wave=%7.1f  => per SIMD %7.1f cycles/iter\n", kN[V], bpc, ms,
           mhz[grid / 2], cyc[grid / 2], cyc[grid / 2] / bpc);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d_out; hipMalloc(&d_out, sizeof(float) * 256 * cus * 4);
    unsigned long long* d_clk; hipMalloc(&d_clk, 16 * cus * 4);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<MFMA_ONLY>(bpc, cus, d_out, d_clk); run<FMA_ONLY>(bpc, cus, d_out, d_clk); run<CVT_ONLY>(bpc, cus, d_out, d_clk);
        run<BOTH_FMA>(bpc, cus, d_out, d_clk); run<BOTH_CVT>(bpc, cus, d_out, d_clk);
        run<LDS_ONLY>(bpc, cus, d_out, d_clk); run<BOTH_LDS>(bpc, cus, d_out, d_clk);
        printf("\n");
    }
    return 0;
}
