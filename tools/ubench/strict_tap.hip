// strict_tap.hip -- the bounded round-2 experiment on the last ~12 % of the strict layer-1+2 kernel.
// One "tap-step" of k_conv12_mfma<strict> = 1 v_mfma_f32_32x32x1_2b_f32 (2048 exact products, C = 0) + 2048 fp32 adds
// + 2 ds_read_b32.  Floor if MFMA and VALU do not overlap: 64 + 64 = 128 cycles per SIMD; the kernel runs at 144-146.
// This measures hand-scheduled inline-asm forms of that step at 2..5 waves per SIMD:
//   PK      16 v_pk_add_f32 after the MFMA (what the compiler emits)
//   PK_PRIO the same with s_setprio 3 around the MFMA issue and 0 for the adds
//   PK_4x4  adds issued as 4 groups of 4 with the next tap's two ds_read_b32 between groups
//   SC      32 v_add_f32 instead of 16 v_pk_add_f32
//   MIX     8 v_pk_add_f32 + 16 v_add_f32
//   PK_ONLY / SC_ONLY / MFMA_ONLY  the parts alone
// All variants keep the real data flow: adds consume the PREVIOUS step's MFMA result (software pipeline of depth 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x32 __attribute__((ext_vector_type(32)));
enum { PK, PK_PRIO, PK_4x4, SC, MIX, PK_ONLY, SC_ONLY, MFMA_ONLY, PK_NP, SC_NP, NV };
const char* kN[NV] = {"mfma + 16 pk_add", "mfma(prio3) + 16 pk_add", "mfma + 4x(4 pk_add) + ds_reads between", "mfma + 32 v_add_f32",
                      "mfma + 8 pk_add + 16 v_add_f32", "16 pk_add only", "32 v_add_f32 only", "mfma only",
                      "mfma -> its own 16 pk_add (one buffer)", "mfma -> its own 32 v_add_f32 (one buffer)"};

#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&acc[2 * (i)]) : "v"(*(const double*)&d[2 * (i)]))
#define SCADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(d[i]))

// one result buffer: the adds consume the MFMA they follow (what variant 1 of the product kernel does); 64 + few VGPRs,
// so up to 6 waves per SIMD fit
template <int V, int WPS>
__global__ __launch_bounds__(256, WPS) void k1(float* out, unsigned long long* clk, int iters, float a0)
{
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 1e-4f;
    __syncthreads();
    float a = a0 + lane * 1e-3f, b = 0.5f + lane * 2e-3f;
    float acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    const f32x32 zero = {};
    const float* lp = lds + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x32 nd = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0);
        float d[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) d[i] = nd[i];
        a = lp[(it & 31) * 64]; b = lp[2048 + (it & 31) * 64];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (V == PK_NP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) PKADD(i);
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) SCADD(i);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a + b;
    for (int i = 0; i < 32; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// software-pipelined forms: the adds consume the PREVIOUS step's products while the next MFMA is in flight.  Two
// explicit product buffers and two steps per loop trip, so the double buffer is a rename, not 32 v_mov.
template <int V, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, unsigned long long* clk, int iters, float a0)
{
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 1e-4f;
    __syncthreads();
    float a = a0 + lane * 1e-3f, b = 0.5f + lane * 2e-3f;
    float acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    const f32x32 zero = {};
    const float* lp = lds + lane;
    f32x32 p0 = zero, p1 = zero;
    for (int i = 0; i < 32; ++i) p1[i] = lane * 1e-6f * i;
    auto half_step = [&](f32x32& produce, const f32x32& consume, int it) {
        float d[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) d[i] = consume[i];
        float na = a, nb = b;
        if constexpr (V != PK_ONLY && V != SC_ONLY) {
            if constexpr (V == PK_PRIO) asm volatile("s_setprio 3");
            produce = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0);
            asm volatile("" : "+v"(produce));
            if constexpr (V == PK_PRIO) asm volatile("s_setprio 0");
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (V == PK || V == PK_PRIO || V == PK_ONLY) {
#pragma unroll
            for (int i = 0; i < 16; ++i) PKADD(i);
            na = lp[(it & 31) * 64]; nb = lp[2048 + (it & 31) * 64];
        } else if constexpr (V == PK_4x4) {
            PKADD(0); PKADD(1); PKADD(2); PKADD(3);
            na = lp[(it & 31) * 64];
            __builtin_amdgcn_sched_barrier(0);
            PKADD(4); PKADD(5); PKADD(6); PKADD(7);
            nb = lp[2048 + (it & 31) * 64];
            __builtin_amdgcn_sched_barrier(0);
            PKADD(8); PKADD(9); PKADD(10); PKADD(11);
            __builtin_amdgcn_sched_barrier(0);
            PKADD(12); PKADD(13); PKADD(14); PKADD(15);
        } else if constexpr (V == SC || V == SC_ONLY) {
#pragma unroll
            for (int i = 0; i < 32; ++i) SCADD(i);
            na = lp[(it & 31) * 64]; nb = lp[2048 + (it & 31) * 64];
        } else if constexpr (V == MIX) {
#pragma unroll
            for (int i = 0; i < 8; ++i) PKADD(i);
#pragma unroll
            for (int i = 16; i < 32; ++i) SCADD(i);
            na = lp[(it & 31) * 64]; nb = lp[2048 + (it & 31) * 64];
        } else {
            na = lp[(it & 31) * 64]; nb = lp[2048 + (it & 31) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        a = na; b = nb;
    };
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
        half_step(p0, p1, it);
        half_step(p1, p0, it + 1);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a + b;
    for (int i = 0; i < 32; ++i) s += acc[i] + p0[i] + p1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int V, int WPS>
void run(int cus, float* d_out, unsigned long long* d_clk)
{
    const int iters = 40000, grid = cus * WPS;
    auto fn = (V == PK_NP || V == SC_NP) ? &k1<V, WPS> : &k<V, WPS>;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, 0, d_out, d_clk, 400, 1.0f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters, 1.0f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(grid);
    (void)hipMemcpy(c.data(), d_clk, grid * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(grid);
    for (int i = 0; i < grid; ++i) cyc[i] = (double)c[i] / iters;
    std::sort(cyc.begin(), cyc.end());
    // Throughput comes from the WALL time of the launch (all blocks resident: grid = CUs x WPS blocks of 4 waves); the
    // per-block s_memtime span is printed only as a cross-check (it under-counts when blocks do not start together).
    const double ns_step = ms * 1e6 / ((double)iters * WPS);
    printf("%-42s waves/SIMD=%d  %7.2f ms  => %6.1f ns per tap-step per SIMD = %6.1f cycles at 2.35 GHz   (s_memtime span/step/wave %7.1f)\n",
           kN[V], WPS, ms, ns_step, ns_step * 2.35, cyc[grid / 2]);
}

template <int WPS>
void all(int cus, float* d_out, unsigned long long* d_clk)
{
    run<MFMA_ONLY, WPS>(cus, d_out, d_clk); run<PK_ONLY, WPS>(cus, d_out, d_clk); run<SC_ONLY, WPS>(cus, d_out, d_clk);
    run<PK, WPS>(cus, d_out, d_clk); run<PK_PRIO, WPS>(cus, d_out, d_clk); run<PK_4x4, WPS>(cus, d_out, d_clk);
    run<SC, WPS>(cus, d_out, d_clk); run<MIX, WPS>(cus, d_out, d_clk);
    run<PK_NP, WPS>(cus, d_out, d_clk); run<SC_NP, WPS>(cus, d_out, d_clk);
    printf("\n");
}

int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d_out; (void)hipMalloc(&d_out, sizeof(float) * 256 * cus * 8);
    unsigned long long* d_clk; (void)hipMalloc(&d_clk, 8 * cus * 8);
    all<2>(cus, d_out, d_clk); all<3>(cus, d_out, d_clk); all<4>(cus, d_out, d_clk);
    run<PK_NP, 5>(cus, d_out, d_clk); run<SC_NP, 5>(cus, d_out, d_clk); run<PK_NP, 6>(cus, d_out, d_clk); run<SC_NP, 6>(cus, d_out, d_clk);
    return 0;
}
