// bf16_split_probe.hip -- can the (16x faster, VALU-concurrent?) bf16 matrix pipe produce correctly rounded
// fp32 products?  w = w1+w2+w3, y = y1+y2+y3 (bf16 pieces, exact), the 9 partial products are laid along K
// of ONE v_mfma_f32_32x32x16_bf16 with C = 0.  If the instruction sums its K terms exactly (wide internal
// accumulator) and rounds once, D == v_mul_f32(w, y) bit for bit.  Also times bf16-MFMA || 16 v_pk_add_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#pragma clang fp contract(off)
#define PIN(v) asm volatile("" : "+v"(v))
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __host__ inline void split3(float v, unsigned short p[3])
{
    unsigned u; memcpy(&u, &v, 4);
    unsigned u1 = u & 0xFFFF0000u; float f1; memcpy(&f1, &u1, 4);
    float r1 = v - f1; unsigned ur; memcpy(&ur, &r1, 4);
    unsigned u2 = ur & 0xFFFF0000u; float f2; memcpy(&f2, &u2, 4);
    float r2 = r1 - f2; unsigned u3; memcpy(&u3, &r2, 4);
    p[0] = u1 >> 16; p[1] = u2 >> 16; p[2] = u3 >> 16;     // r2 has <= 8 significant bits -> exact bf16
}

// order of the 9 (i,j) partial products along k: ascending magnitude or descending
__global__ void probe(const float* w, const float* y, float* d_mfma, float* d_mul, int order)
{
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    unsigned short wp[3], yp[3];
    split3(w[r], wp);          // A row r
    split3(y[r], yp);          // B col r
    // k -> (i,j)
    int ki[16], kj[16];
    for (int k = 0; k < 16; ++k) { ki[k] = -1; kj[k] = -1; }
    int n = 0;
    if (order == 0) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { ki[n] = i; kj[n] = j; ++n; } }
    else { for (int s = 4; s >= 0; --s) for (int i = 0; i < 3; ++i) { int j = s - i; if (j >= 0 && j < 3) { ki[n] = i; kj[n] = j; ++n; } } }
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * h + e;
        a[e] = ki[k] >= 0 ? (short)wp[ki[k]] : (short)0;
        b[e] = kj[k] >= 0 ? (short)yp[kj[k]] : (short)0;
    }
    f32x16 z = {};
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
    for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
        d_mfma[row * 32 + r] = d[q];
        d_mul[row * 32 + r] = w[row] * y[r];
    }
}

template <int V>   // 0: bf16 mfma only, 1: + 8 pk_add (the 32x32 tile = 16 regs), 2: 8 pk_add only,
                   // 3: 32 independent v_add_f32 only, 4: bf16 mfma + those 32 adds (throughput-bound co-issue test)
__global__ __launch_bounds__(256, 2) void tim(float* out, unsigned long long* clk, int iters)
{
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(0x3f80 + lane + e); b[e] = (short)(0x3f00 + lane * 3 + e); }
    f32x16 z = {}, acc = {}, d_cur = {};
    float wide[32]; float inc = 1e-7f * lane;
    for (int i = 0; i < 32; ++i) wide[i] = i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (V == 0) {
            f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0); PIN(d);
            __builtin_amdgcn_sched_barrier(0);
            d_cur = d;
        } else if constexpr (V == 1) {
            f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0); PIN(d);
            __builtin_amdgcn_sched_barrier(0);
            acc += d_cur; PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
            d_cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, z, 0, 0, 0); PIN(d_cur);
            __builtin_amdgcn_sched_barrier(0);
            acc += d; PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (V == 2) {
            acc += d_cur; PIN(acc);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            if constexpr (V == 4) {
                f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0); PIN(d);
                __builtin_amdgcn_sched_barrier(0);
                d_cur = d;
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(wide[i]) : "v"(inc));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i] + d_cur[i];
    for (int i = 0; i < 32; ++i) s += wide[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main()
{
    float *dw, *dy, *dm, *dv;
    hipMalloc(&dw, 128); hipMalloc(&dy, 128); hipMalloc(&dm, 4096); hipMalloc(&dv, 4096);
    srand(1);
    for (int order = 0; order < 2; ++order) {
        long bad = 0, total = 0; double maxulp = 0;
        for (int trial = 0; trial < 2000; ++trial) {
            float w[32], y[32];
            for (int i = 0; i < 32; ++i) {
                w[i] = ((rand() % 20001) - 10000) * 1e-4f * ((trial & 1) ? 1.f : (float)(rand() % 1000 + 1) / 777.f);
                y[i] = (float)rand() / RAND_MAX * 255.f * ((trial & 2) ? 1.f : 1e-3f) + ((trial & 4) ? 0.f : -3.f);
            }
            hipMemcpy(dw, w, 128, hipMemcpyHostToDevice); hipMemcpy(dy, y, 128, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dw, dy, dm, dv, order);
            float m[1024], v[1024];
            hipMemcpy(m, dm, 4096, hipMemcpyDeviceToHost); hipMemcpy(v, dv, 4096, hipMemcpyDeviceToHost);
            for (int i = 0; i < 1024; ++i) {
                ++total;
                if (memcmp(&m[i], &v[i], 4) != 0) {
                    ++bad;
                    double ulp = fabs((double)m[i] - (double)v[i]) / (fabs((double)v[i]) * 5.96e-8 + 1e-300);
                    if (ulp > maxulp) maxulp = ulp;
                    if (bad <= 3) printf("  mismatch: w*y via bf16-split MFMA = %.9g, v_mul_f32 = %.9g\n", m[i], v[i]);
                }
            }
        }
        printf("order %s: %ld / %ld products differ from v_mul_f32 (max %.2f ulp)\n", order ? "descending-magnitude" : "i-major", bad, total, maxulp);
    }
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d_out; hipMalloc(&d_out, sizeof(float) * 256 * cus * 2);
    unsigned long long* d_clk; hipMalloc(&d_clk, 8 * cus * 2);
    const int iters = 100000;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int grid = cus * bpc;
        auto report = [&](const char* name, int steps, int nadd) {
            std::vector<unsigned long long> c(grid);
            hipDeviceSynchronize();
            hipMemcpy(c.data(), d_clk, grid * 8, hipMemcpyDeviceToHost);
            printf("%-28s waves/SIMD=%d: %.1f cycles per step per wave => %.1f cycles per step per SIMD  (step = 1 mfma_32x32x16_bf16 %s)\n",
                   name, bpc, (double)c[grid / 2] / iters / steps, (double)c[grid / 2] / iters / steps / bpc, nadd ? "+ 8 v_pk_add_f32" : "");
        };
        hipLaunchKernelGGL(tim<0>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters); report("bf16 mfma only", 1, 0);
        hipLaunchKernelGGL(tim<1>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters); report("bf16 mfma + 8 pk_add", 2, 1);
        hipLaunchKernelGGL(tim<2>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters); report("8 pk_add only", 1, 1);
        hipLaunchKernelGGL(tim<3>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters); report("32 v_add_f32 only", 1, 1);
        hipLaunchKernelGGL(tim<4>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters); report("bf16 mfma + 32 v_add_f32", 1, 1);
    }
    return 0;
}
