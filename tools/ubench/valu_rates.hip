// valu_rates.hip -- VALU issue-rate microbenchmark for gfx950 (MI355X).
// Answers the design questions for the strict (no-FMA) SRCNN kernels:
//   * is v_pk_mul_f32 / v_pk_add_f32 faster per flop than v_mul_f32 / v_add_f32 ?
//   * what do v_cvt_f64_f32 and v_add_f64 cost (conv 5x5 accumulates fp32 products in fp64) ?
// Each thread runs CH independent dependency chains for `iters` rounds; results are reported as
// wave-instructions per cycle per SIMD assuming the clock given on the command line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#pragma clang fp contract(off)

typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int CH = 16;

enum Mode { FMA32, MULADD32, PKFMA, PKMULADD, MUL_CVT_ADD64, ADD64, FMA64, CVT64, MULADD32_SGPR, PKMULADD_SGPR, MUL64, MULADD64, FMA0ADD64, PKMUL_CVT_ADD64, CVT_ADD64, CVT_F64_F32, NMODES };
const char* kNames[NMODES] = {"v_fma_f32", "v_mul_f32+v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32+v_pk_add_f32",
                              "v_mul_f32+v_cvt_f64_f32+v_add_f64", "v_add_f64", "v_fma_f64", "v_cvt_f64_f32+v_cvt_f32_f64",
                              "v_mul_f32(sgpr)+v_add_f32", "v_pk_mul_f32(sgpr)+v_pk_add_f32",
                              "v_mul_f64", "v_cvt_f64_f32+v_mul_f64+v_add_f64 (resampler tap)", "v_cvt_f64_f32+v_fma_f64(w,x,0)+v_add_f64",
                              "v_pk_mul_f32 + 2 v_cvt_f64_f32 + 2 v_add_f64 (conv3 pair)", "v_cvt_f64_f32+v_add_f64", "v_cvt_f64_f32"};
// VALU instructions per chain step, and useful "MAC-equivalents" (multiply-accumulates) per chain step
const int kInstr[NMODES] = {1, 2, 1, 2, 3, 1, 1, 2, 2, 2, 1, 3, 3, 5, 2, 1};
const int kMacs[NMODES]  = {1, 1, 2, 2, 1, 1, 1, 1, 1, 2, 1, 1, 1, 2, 1, 1};

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters, const float* __restrict__ wt)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (MODE == FMA32) {
        float x[CH];
        for (int c = 0; c < CH; ++c) x[c] = tid * 1e-9f + c;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) x[c] = __builtin_fmaf(x[c], a, b);
        float s = 0; for (int c = 0; c < CH; ++c) s += x[c];
        out[tid] = s;
    } else if constexpr (MODE == MULADD32) {
        float x[CH];
        for (int c = 0; c < CH; ++c) x[c] = tid * 1e-9f + c;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) { float t = x[c] * a; x[c] = t + b; }
        float s = 0; for (int c = 0; c < CH; ++c) s += x[c];
        out[tid] = s;
    } else if constexpr (MODE == PKFMA) {
        float2v x[CH]; float2v av = {a, a * 1.0001f}, bv = {b, b * 0.999f};
        for (int c = 0; c < CH; ++c) x[c] = float2v{tid * 1e-9f + c, tid * 2e-9f + c};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) x[c] = __builtin_elementwise_fma(x[c], av, bv);
        float s = 0; for (int c = 0; c < CH; ++c) s += x[c].x + x[c].y;
        out[tid] = s;
    } else if constexpr (MODE == PKMULADD) {
        float2v x[CH]; float2v av = {a, a * 1.0001f}, bv = {b, b * 0.999f};
        for (int c = 0; c < CH; ++c) x[c] = float2v{tid * 1e-9f + c, tid * 2e-9f + c};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) { float2v t = x[c] * av; x[c] = t + bv; }
        float s = 0; for (int c = 0; c < CH; ++c) s += x[c].x + x[c].y;
        out[tid] = s;
    } else if constexpr (MODE == MUL_CVT_ADD64) {
        double acc[CH]; float v[CH];
        for (int c = 0; c < CH; ++c) { acc[c] = tid * 1e-9 + c; v[c] = tid * 1e-9f + c; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) { float p = v[c] * a; acc[c] = acc[c] + (double)p; v[c] = p; }
        double s = 0; for (int c = 0; c < CH; ++c) s += acc[c];
        out[tid] = (float)s;
    } else if constexpr (MODE == PKMUL_CVT_ADD64) {
        // conv3's inner step as the kernel issues it: one packed product for two taps, two conversions, two fp64 adds
        double acc[CH], acc2[CH]; float2v v[CH]; const float2v av = {a, a * 1.0001f};
        for (int c = 0; c < CH; ++c) { acc[c] = tid * 1e-9 + c; acc2[c] = c; v[c] = float2v{tid * 1e-9f + c, tid * 2e-9f + c}; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) { float2v p = v[c] * av; acc[c] = acc[c] + (double)p.x; acc2[c] = acc2[c] + (double)p.y; v[c] = p; }
        double s = 0; for (int c = 0; c < CH; ++c) s += acc[c] + acc2[c];
        out[tid] = (float)s;
    } else if constexpr (MODE == CVT_ADD64) {
        double acc[CH]; float v[CH];
        for (int c = 0; c < CH; ++c) { acc[c] = tid * 1e-9 + c; v[c] = tid * 1e-9f + c; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) { float x = v[c]; asm volatile("" : "+v"(x)); acc[c] = acc[c] + (double)x; }
        double s = 0; for (int c = 0; c < CH; ++c) s += acc[c];
        out[tid] = (float)s;
    } else if constexpr (MODE == CVT_F64_F32) {
        float v[CH]; double d[CH];
        for (int c = 0; c < CH; ++c) { v[c] = tid * 1e-9f + c; d[c] = 0; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) { float x = v[c]; asm volatile("" : "+v"(x)); d[c] = (double)x; asm volatile("" : "+v"(d[c])); }
        double s = 0; for (int c = 0; c < CH; ++c) s += d[c];
        out[tid] = (float)s;
    } else if constexpr (MODE == ADD64) {
        double acc[CH]; const double bd = b;
        for (int c = 0; c < CH; ++c) acc[c] = tid * 1e-9 + c;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = acc[c] + bd;
        double s = 0; for (int c = 0; c < CH; ++c) s += acc[c];
        out[tid] = (float)s;
    } else if constexpr (MODE == FMA64) {
        double acc[CH]; const double ad = a, bd = b;
        for (int c = 0; c < CH; ++c) acc[c] = tid * 1e-9 + c;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_fma(acc[c], ad, bd);
        double s = 0; for (int c = 0; c < CH; ++c) s += acc[c];
        out[tid] = (float)s;
    } else if constexpr (MODE == MUL64) {
        double acc[CH]; const double ad = 1.0 + 1e-9 * a;
        for (int c = 0; c < CH; ++c) acc[c] = tid * 1e-9 + c + 1;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = acc[c] * ad;
        double s = 0; for (int c = 0; c < CH; ++c) s += acc[c];
        out[tid] = (float)s;
    } else if constexpr (MODE == MULADD64 || MODE == FMA0ADD64) {
        // the resampler's tap: acc = acc + w * (double)x, product and sum rounded separately.  FMA0ADD64 writes the product
        // as fma(w, x, +0.0): the same value (the +0.0 only matters for a -0 product, which the add swallows either way)
        double acc[CH]; float v[CH]; const double wd = 0.25 + 1e-9 * a;
        for (int c = 0; c < CH; ++c) { acc[c] = tid * 1e-9 + c; v[c] = tid * 1e-9f + c; }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const double x = (double)v[c];
                double p;
                if constexpr (MODE == MULADD64) p = wd * x; else p = __builtin_fma(wd, x, 0.0);
                acc[c] = acc[c] + p;
                v[c] = v[c] + 1.0f;
            }
        double s = 0; for (int c = 0; c < CH; ++c) s += acc[c];
        out[tid] = (float)s;
    } else if constexpr (MODE == CVT64) {
        float v[CH];
        for (int c = 0; c < CH; ++c) v[c] = tid * 1e-9f + c;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < CH; ++c) { double d = (double)v[c]; asm volatile("" : "+v"(d)); v[c] = (float)d; }
        float s = 0; for (int c = 0; c < CH; ++c) s += v[c];
        out[tid] = s;
    } else if constexpr (MODE == MULADD32_SGPR) {
        // the conv-1 shape: accumulator += (uniform weight from memory) * (per-lane value)
        float x[CH]; float y = tid * 1e-9f + a;
        for (int c = 0; c < CH; ++c) x[c] = c;
        for (int it = 0; it < iters; ++it) {
            const float* w = wt + (it & 63) * CH;
#pragma unroll
            for (int c = 0; c < CH; ++c) { float t = w[c] * y; x[c] = x[c] + t; }
        }
        float s = 0; for (int c = 0; c < CH; ++c) s += x[c];
        out[tid] = s;
    } else if constexpr (MODE == PKMULADD_SGPR) {
        float2v x[CH]; float2v y = {tid * 1e-9f + a, tid * 2e-9f + b};
        for (int c = 0; c < CH; ++c) x[c] = float2v{(float)c, (float)c};
        for (int it = 0; it < iters; ++it) {
            const float* w = wt + (it & 63) * CH;
#pragma unroll
            for (int c = 0; c < CH; ++c) { float2v t = y * w[c]; x[c] = x[c] + t; }
        }
        float s = 0; for (int c = 0; c < CH; ++c) s += x[c].x + x[c].y;
        out[tid] = s;
    }
}

template <int MODE>
void run(int blocks_per_cu, int cus, double ghz, float* d_out, const float* d_w)
{
    const int iters = 20000;
    const int grid = cus * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 1.0000001f, 1e-7f, 100, d_w);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 1.0000001f, 1e-7f, iters, d_w);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double steps = (double)grid * 256 * iters * CH;            // lane chain-steps
    const double wave_instr = steps / 64 * kInstr[MODE];
    const double per_simd_cycle = wave_instr / (best * 1e-3) / (cus * 4.0) / (ghz * 1e9);
    const double tmacs = steps * kMacs[MODE] / (best * 1e-3) / 1e12;
    printf("%-40s blocks/CU=%d  %8.3f ms  %6.3f wave-instr/cycle/SIMD (=%5.2f cyc/instr)  %7.2f TMAC/s\n",
           kNames[MODE], blocks_per_cu, best, per_simd_cycle, 1.0 / per_simd_cycle, tmacs);
}

int main(int argc, char** argv)
{
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("device %s %s CUs=%d clock(assumed)=%.2f GHz  clockRate=%d kHz\n", p.name, p.gcnArchName, cus, ghz, p.clockRate);
    float* d_out; hipMalloc(&d_out, sizeof(float) * 256 * cus * 8);
    std::vector<float> w(64 * CH); for (size_t i = 0; i < w.size(); ++i) w[i] = 1e-3f * (float)(i % 17);
    float* d_w; hipMalloc(&d_w, w.size() * 4); hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    for (int b : {1, 2, 4}) {
        run<FMA32>(b, cus, ghz, d_out, d_w);
        run<MULADD32>(b, cus, ghz, d_out, d_w);
        run<PKFMA>(b, cus, ghz, d_out, d_w);
        run<PKMULADD>(b, cus, ghz, d_out, d_w);
        run<MUL_CVT_ADD64>(b, cus, ghz, d_out, d_w);
        run<ADD64>(b, cus, ghz, d_out, d_w);
        run<FMA64>(b, cus, ghz, d_out, d_w);
        run<CVT64>(b, cus, ghz, d_out, d_w);
        run<MULADD32_SGPR>(b, cus, ghz, d_out, d_w);
        run<PKMULADD_SGPR>(b, cus, ghz, d_out, d_w);
        run<MUL64>(b, cus, ghz, d_out, d_w);
        run<MULADD64>(b, cus, ghz, d_out, d_w);
        run<FMA0ADD64>(b, cus, ghz, d_out, d_w);
        run<PKMUL_CVT_ADD64>(b, cus, ghz, d_out, d_w);
        run<CVT_ADD64>(b, cus, ghz, d_out, d_w);
        run<CVT_F64_F32>(b, cus, ghz, d_out, d_w);
        printf("\n");
    }
    return 0;
}
