// mfma_coissue.hip -- how well do the matrix pipe and the VALU overlap on gfx950 for the strict
// conv kernel's inner pattern:   D = mfma_32x32x1_2b(a, b, 0)  ||  acc += D_prev (16 v_pk_add_f32) ?
// Variants: MFMA only, adds only, both (software-pipelined exactly like k_conv12_mfma).
// Also reports the in-kernel shader clock: d(s_memtime) / d(s_memrealtime) * 100 MHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#pragma clang fp contract(off)
#define PIN(v) asm volatile("" : "+v"(v))
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { MFMA_ONLY, ADDS_ONLY, BOTH, BOTH_ADDS8, BOTH_ADDS12, MFMA16_BOTH, SCALAR_ONLY, BOTH_SCALAR, F64_ONLY, BOTH_F64, CVT_ONLY, BOTH_CVT, NV };
const char* kN[NV] = {"mfma32x32x1_2b only", "16 pk_add only", "mfma32 + 16 pk_add", "mfma32 + 8 pk_add", "mfma32 + 12 pk_add",
                      "mfma16x16x1_4b + 8 pk_add", "32 v_add_f32 only", "mfma32 + 32 v_add_f32", "16 v_add_f64 only", "mfma32 + 16 v_add_f64", "16 v_cvt_f64_f32 only", "mfma32 + 16 v_cvt_f64_f32"};

template <int V>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* clk, int iters, float a0, float b0)
{
    const int lane = threadIdx.x & 63;
    float a = a0 + lane * 1e-3f, b = b0 + lane * 2e-3f;
    const f32x32 zero = {};
    f32x32 acc = zero, d_cur = zero;
    double dacc[16]; double dinc = 1e-9 * lane;
    for (int i = 0; i < 16; ++i) dacc[i] = i;
    for (int i = 0; i < 32; ++i) d_cur[i] = lane * 1e-6f * i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == MFMA_ONLY) {
            f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0); PIN(d);
            __builtin_amdgcn_sched_barrier(0);
            d_cur = d;
        } else if constexpr (V == ADDS_ONLY) {
            acc += d_cur; PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (V == BOTH) {
            // two steps per trip so the double buffer is a rename, not 32 v_mov
            f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0); PIN(d);
            __builtin_amdgcn_sched_barrier(0);
            acc += d_cur; PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
            d_cur = __builtin_amdgcn_mfma_f32_32x32x1f32(b, a, zero, 0, 0, 0); PIN(d_cur);
            __builtin_amdgcn_sched_barrier(0);
            acc += d; PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (V == BOTH_ADDS8 || V == BOTH_ADDS12) {
            constexpr int NA = (V == BOTH_ADDS8) ? 16 : 24;
            f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0); PIN(d);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] += d_cur[i];
            PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
            d_cur = __builtin_amdgcn_mfma_f32_32x32x1f32(b, a, zero, 0, 0, 0); PIN(d_cur);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i] += d[i];
            PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (V == MFMA16_BOTH) {
            const f32x16 z16 = {};
            f32x16 d = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, z16, 0, 0, 0); PIN(d);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += d_cur[i];
            PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 e = __builtin_amdgcn_mfma_f32_16x16x1f32(b, a, z16, 0, 0, 0); PIN(e);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += d[i];
            PIN(acc);
#pragma unroll
            for (int i = 0; i < 16; ++i) d_cur[i] = e[i];
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (V == SCALAR_ONLY) {
#pragma unroll
            for (int i = 0; i < 32; ++i) { float x = acc[i]; asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(d_cur[i])); acc[i] = x; }
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (V == BOTH_SCALAR) {
            f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0); PIN(d);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 32; ++i) { float x = acc[i]; asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(d_cur[i])); acc[i] = x; }
            __builtin_amdgcn_sched_barrier(0);
            d_cur = __builtin_amdgcn_mfma_f32_32x32x1f32(b, a, zero, 0, 0, 0); PIN(d_cur);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 32; ++i) { float x = acc[i]; asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(d[i])); acc[i] = x; }
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (V == F64_ONLY || V == BOTH_F64) {
            // 16 independent fp64 adds on registers that the MFMA never touches
            if constexpr (V == BOTH_F64) {
                f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0); PIN(d);
                __builtin_amdgcn_sched_barrier(0);
                d_cur = d;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dacc[i]) : "v"(dinc));
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (V == CVT_ONLY || V == BOTH_CVT) {
            if constexpr (V == BOTH_CVT) {
                f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0); PIN(d);
                __builtin_amdgcn_sched_barrier(0);
                d_cur = d;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dacc[i]) : "v"(a));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 32; ++i) s += acc[i] + d_cur[i];
    for (int i = 0; i < 16; ++i) s += (float)dacc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int V>
void run(int bpc, int cus, float* d_out, unsigned long long* d_clk)
{
    const int iters = 100000, grid = cus * bpc;
    const int steps = (V == BOTH || V == BOTH_ADDS8 || V == BOTH_ADDS12 || V == MFMA16_BOTH || V == BOTH_SCALAR) ? 2 : 1;
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, 1000, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(2 * grid);
    hipMemcpy(c.data(), d_clk, c.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> mhz(grid), cyc(grid);
    for (int i = 0; i < grid; ++i) { mhz[i] = (double)c[2 * i] / (double)c[2 * i + 1] * 100.0; cyc[i] = (double)c[2 * i] / iters / steps; }
    std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
    // iterations per SIMD: bpc waves per SIMD each doing `iters`
    const double ns_per_iter_simd = ms * 1e6 / ((double)iters * steps * bpc);
    printf("%-30s waves/SIMD=%d  %8.2f ms  clock(median)=%6.0f MHz  shader-cycles/iter/wave=%7.1f  => per SIMD: %6.1f ns/iter = %6.1f cycles/iter\n",
           kN[V], bpc, ms, mhz[grid / 2], cyc[grid / 2], ns_per_iter_simd, ns_per_iter_simd * mhz[grid / 2] * 1e-3);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d_out; hipMalloc(&d_out, sizeof(float) * 256 * cus * 4);
    unsigned long long* d_clk; hipMalloc(&d_clk, 16 * cus * 4);
    for (int bpc : {1, 2}) {
        run<MFMA_ONLY>(bpc, cus, d_out, d_clk);
        run<ADDS_ONLY>(bpc, cus, d_out, d_clk);
        run<BOTH>(bpc, cus, d_out, d_clk);
        run<BOTH_ADDS12>(bpc, cus, d_out, d_clk);
        run<BOTH_ADDS8>(bpc, cus, d_out, d_clk);
        run<MFMA16_BOTH>(bpc, cus, d_out, d_clk);
        run<SCALAR_ONLY>(bpc, cus, d_out, d_clk);
        run<BOTH_SCALAR>(bpc, cus, d_out, d_clk);
        run<F64_ONLY>(bpc, cus, d_out, d_clk);
        run<BOTH_F64>(bpc, cus, d_out, d_clk);
        run<CVT_ONLY>(bpc, cus, d_out, d_clk);
        run<BOTH_CVT>(bpc, cus, d_out, d_clk);
        printf("\n");
    }
    return 0;
}
