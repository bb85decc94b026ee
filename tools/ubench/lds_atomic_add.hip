// lds_atomic_add.hip -- could the LDS float-atomic unit take a share of the strict accumulate (acc += D)
// concurrently with the VALU/MFMA?  Measures ds_add_f32 alone, beside the fp32 MFMA + 16 packed adds, and checks
// its arithmetic: round-to-nearest-even?  denormals preserved?  (bit-compared with v_add_f32 on the same data).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#pragma clang fp contract(off)
#define PIN(v) asm volatile("" : "+v"(v))
typedef float f32x32 __attribute__((ext_vector_type(32)));

__global__ void semantics(const float* a, const float* b, float* lds_res, float* valu_res, int n)
{
    __shared__ float acc[256];
    const int t = threadIdx.x;
    for (int i = t; i < n; i += 256) {
        acc[t] = a[i];
        __builtin_amdgcn_s_waitcnt(0);
        __hip_atomic_fetch_add(&acc[t], b[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_s_waitcnt(0);
        lds_res[i] = acc[t];
        valu_res[i] = a[i] + b[i];
    }
}

template <int V>   // 0: 4 ds_add_f32 only; 1: mfma + 16 pk_add; 2: mfma + 14 pk_add + 4 ds_add_f32 (same 2048 adds)
__global__ __launch_bounds__(256, 2) void tim(float* out, unsigned long long* clk, int iters, float a0, float b0)
{
    __shared__ float lacc[4 * 4 * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* mine = lacc + wv * 256 + lane;
    for (int r = 0; r < 4; ++r) mine[r * 64] = 0.f;
    float a = a0 + lane * 1e-3f, b = b0 + lane * 2e-3f;
    const f32x32 zero = {};
    f32x32 acc = zero, d_cur = zero;
    for (int i = 0; i < 32; ++i) d_cur[i] = lane * 1e-6f * i;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) __hip_atomic_fetch_add(mine + r * 64, d_cur[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (V == 1) {
            f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0); PIN(d);
            __builtin_amdgcn_sched_barrier(0);
            acc += d; PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            f32x32 d = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, zero, 0, 0, 0); PIN(d);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) __hip_atomic_fetch_add(mine + r * 64, d[28 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int r = 0; r < 28; ++r) acc[r] += d[r];
            PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 32; ++i) s += acc[i];
    for (int r = 0; r < 4; ++r) s += mine[r * 64];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main()
{
    // ---- semantics ----
    const int n = 1 << 16;
    std::vector<float> a(n), b(n);
    srand(7);
    for (int i = 0; i < n; ++i) {
        unsigned ua = ((unsigned)rand() << 16) ^ (unsigned)rand(), ub = ((unsigned)rand() << 16) ^ (unsigned)rand();
        if (i % 4 == 0) { ua &= 0x807fffffu; }                 // denormal a
        if (i % 4 == 1) { ub &= 0x807fffffu; ua &= 0x80ffffffu; }   // denormal b, tiny a
        if (i % 8 == 7) { ub = ua ^ 0x80000001u; }             // near cancellation -> denormal/zero results
        memcpy(&a[i], &ua, 4); memcpy(&b[i], &ub, 4);
        if (a[i] != a[i] || b[i] != b[i]) { a[i] = 1.5f; b[i] = 2.5f; }
    }
    float *da, *db, *dl, *dv;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dl, n * 4); hipMalloc(&dv, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(semantics, dim3(1), dim3(256), 0, 0, da, db, dl, dv, n);
    std::vector<float> l(n), v(n);
    hipMemcpy(l.data(), dl, n * 4, hipMemcpyDeviceToHost); hipMemcpy(v.data(), dv, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad_denorm = 0;
    for (int i = 0; i < n; ++i) {
        if (memcmp(&l[i], &v[i], 4) != 0 && !(l[i] != l[i] && v[i] != v[i])) {
            ++bad;
            unsigned uv; memcpy(&uv, &v[i], 4);
            if ((uv & 0x7f800000u) == 0) ++bad_denorm;
            if (bad <= 4) printf("  differ: a=%.9g b=%.9g  ds_add_f32=%.9g  v_add_f32=%.9g\n", a[i], b[i], l[i], v[i]);
        }
    }
    printf("ds_add_f32 vs v_add_f32 on %d pairs (incl. denormal operands/results): %ld differ (%ld of them where the VALU result is denormal/zero)\n", n, bad, bad_denorm);

    // ---- timing ----
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, iters = 100000;
    float* d_out; hipMalloc(&d_out, sizeof(float) * 256 * cus * 2);
    unsigned long long* d_clk; hipMalloc(&d_clk, 8 * cus * 2);
    const char* names[3] = {"4 ds_add_f32 only", "mfma32 + 16 pk_add", "mfma32 + 14 pk_add + 4 ds_add_f32"};
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int grid = cus * bpc;
        for (int vv = 0; vv < 3; ++vv) {
            if (vv == 0) hipLaunchKernelGGL(tim<0>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters, 1.f, .5f);
            if (vv == 1) hipLaunchKernelGGL(tim<1>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters, 1.f, .5f);
            if (vv == 2) hipLaunchKernelGGL(tim<2>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters, 1.f, .5f);
            hipDeviceSynchronize();
            std::vector<unsigned long long> c(grid);
            hipMemcpy(c.data(), d_clk, grid * 8, hipMemcpyDeviceToHost);
            std::sort(c.begin(), c.end());
            printf("%-36s waves/SIMD=%d: %.1f shader cycles per iteration per wave => %.1f per SIMD\n", names[vv], bpc,
                   (double)c[grid / 2] / iters, (double)c[grid / 2] / iters / bpc);
        }
    }
    return 0;
}
