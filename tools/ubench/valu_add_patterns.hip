// valu_add_patterns.hip -- round 5: WHY do the 2048 fp32 adds of a strict tap-step cost 80 cycles instead of 64?
// (profiles/r01_mfma_coissue.txt: 16 v_pk_add_f32 = 78-89 cycles, 32 v_add_f32 = 80.5; the ideal is 64.)
// Register numbers are explicit, so the patterns differ ONLY in which VGPR banks (register index mod 4) the operands sit in,
// whether the destination is one of the sources, and whether one source is an SGPR.  2048 adds per iteration in every pattern.
//   PK_SAME   16 v_pk_add_f32 acc += d, acc and d on the same bank pair        PK_OTHER  d on the other bank pair
//   PK_PING   16 v_pk_add_f32 acc' = acc + d (destination is a third register range; ranges swap every iteration)
//   PK_SGPR   16 v_pk_add_f32 acc += s[..] (one source scalar: half the vector-register reads)
//   SC_0..3   32 v_add_f32 acc[i] += d[i + k]: source on bank offset k
//   SC_SGPR   32 v_add_f32 acc += s
// Waves per SIMD 1..6 (blocks of 256 threads = one wave per SIMD each).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

enum { PK_SAME, PK_OTHER, PK_PING, PK_SGPR, SC_0, SC_1, SC_2, SC_3, SC_SGPR, NPAT };
const char* kName[NPAT] = {"16 pk_add, d on the same bank pair", "16 pk_add, d on the other bank pair", "16 pk_add, ping-pong destination",
                           "16 pk_add, scalar source", "32 v_add_f32, d bank +0", "32 v_add_f32, d bank +1", "32 v_add_f32, d bank +2",
                           "32 v_add_f32, d bank +3", "32 v_add_f32, scalar source"};

// acc = v[8..39], d = v[40..75], spare = v[76..107]
#define PK(dst, a, b) "v_pk_add_f32 v[" #dst ":" #dst "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1]\n\t"
#define PK16(D, A, B) PK(D+0, A+0, B+0) PK(D+2, A+2, B+2) PK(D+4, A+4, B+4) PK(D+6, A+6, B+6) PK(D+8, A+8, B+8) PK(D+10, A+10, B+10) PK(D+12, A+12, B+12) \
    PK(D+14, A+14, B+14) PK(D+16, A+16, B+16) PK(D+18, A+18, B+18) PK(D+20, A+20, B+20) PK(D+22, A+22, B+22) PK(D+24, A+24, B+24) PK(D+26, A+26, B+26) \
    PK(D+28, A+28, B+28) PK(D+30, A+30, B+30)
#define PKS(dst) "v_pk_add_f32 v[" #dst ":" #dst "+1], v[" #dst ":" #dst "+1], s[20:21]\n\t"
#define PKS16 PKS(8) PKS(10) PKS(12) PKS(14) PKS(16) PKS(18) PKS(20) PKS(22) PKS(24) PKS(26) PKS(28) PKS(30) PKS(32) PKS(34) PKS(36) PKS(38)
#define SC(a, b) "v_add_f32 v[" #a "], v[" #a "], v[" #b "]\n\t"
#define SC8(A, B) SC(A+0, B+0) SC(A+1, B+1) SC(A+2, B+2) SC(A+3, B+3) SC(A+4, B+4) SC(A+5, B+5) SC(A+6, B+6) SC(A+7, B+7)
#define SC32(B) SC8(8, B) SC8(16, B+8) SC8(24, B+16) SC8(32, B+24)
#define SCS(a) "v_add_f32 v[" #a "], s20, v[" #a "]\n\t"
#define SCS8(A) SCS(A+0) SCS(A+1) SCS(A+2) SCS(A+3) SCS(A+4) SCS(A+5) SCS(A+6) SCS(A+7)
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \
    "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75", \
    "v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","s20","s21"

template <int P, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, unsigned long long* clk, int iters)
{
    asm volatile("s_mov_b32 s20, 0x3a83126f\n\ts_mov_b32 s21, 0x3a83126f" ::: "s20", "s21");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 2) {
        if constexpr (P == PK_SAME)       { asm volatile(PK16(8, 8, 40) ::: CLOB); asm volatile(PK16(8, 8, 40) ::: CLOB); }
        else if constexpr (P == PK_OTHER) { asm volatile(PK16(8, 8, 42) ::: CLOB); asm volatile(PK16(8, 8, 42) ::: CLOB); }
        else if constexpr (P == PK_PING)  { asm volatile(PK16(76, 8, 40) ::: CLOB); asm volatile(PK16(8, 76, 40) ::: CLOB); }
        else if constexpr (P == PK_SGPR)  { asm volatile(PKS16 ::: CLOB); asm volatile(PKS16 ::: CLOB); }
        else if constexpr (P == SC_0)     { asm volatile(SC32(40) ::: CLOB); asm volatile(SC32(40) ::: CLOB); }
        else if constexpr (P == SC_1)     { asm volatile(SC32(41) ::: CLOB); asm volatile(SC32(41) ::: CLOB); }
        else if constexpr (P == SC_2)     { asm volatile(SC32(42) ::: CLOB); asm volatile(SC32(42) ::: CLOB); }
        else if constexpr (P == SC_3)     { asm volatile(SC32(43) ::: CLOB); asm volatile(SC32(43) ::: CLOB); }
        else                              { asm volatile(SCS8(8) SCS8(16) SCS8(24) SCS8(32) ::: CLOB); asm volatile(SCS8(8) SCS8(16) SCS8(24) SCS8(32) ::: CLOB); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float v; asm volatile("v_mov_b32 %0, v8" : "=v"(v));
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int P, int WPS>
void run(int cus, float* d_out, unsigned long long* d_clk)
{
    const int iters = 40000, grid = cus * WPS;
    hipLaunchKernelGGL((k<P, WPS>), dim3(grid), dim3(256), 0, 0, d_out, d_clk, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<P, WPS>), dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(2 * grid);
    hipMemcpy(c.data(), d_clk, grid * 16, hipMemcpyDeviceToHost);
    std::vector<double> mhz(grid), cyc(grid);
    for (int i = 0; i < grid; ++i) { mhz[i] = (double)c[2 * i] / (double)c[2 * i + 1] * 100.0; cyc[i] = (double)c[2 * i] / iters; }
    std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
    // WALL-TIME based: the grid holds WPS waves per SIMD on average (the dispatcher need not spread them evenly, so per-wave
    // cycle counts divided by WPS would lie); every SIMD executes iters * WPS iterations of 2048 adds in `ms`
    const double ns_per_iter = ms * 1e6 / ((double)iters * WPS);
    printf("%-40s waves/SIMD=%d  %.2f ms  clock %.0f MHz  per SIMD %.1f ns = %6.1f cycles per 2048 adds   (a wave's own iteration: median %.1f, max %.1f cycles)\n",
           kName[P], WPS, ms, mhz[grid / 2], ns_per_iter, ns_per_iter * mhz[grid / 2] * 1e-3, cyc[grid / 2], cyc[grid - 1]);
}

template <int WPS>
void all(int cus, float* o, unsigned long long* c)
{
    run<PK_SAME, WPS>(cus, o, c); run<PK_OTHER, WPS>(cus, o, c); run<PK_PING, WPS>(cus, o, c); run<PK_SGPR, WPS>(cus, o, c);
    run<SC_0, WPS>(cus, o, c); run<SC_1, WPS>(cus, o, c); run<SC_2, WPS>(cus, o, c); run<SC_3, WPS>(cus, o, c); run<SC_SGPR, WPS>(cus, o, c);
    printf("\n");
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d_out; hipMalloc(&d_out, sizeof(float) * 256 * cus * 8);
    unsigned long long* d_clk; hipMalloc(&d_clk, 16 * cus * 8);
    printf("%s, %d CUs; ideal 64 cycles per 2048 adds per SIMD (32 lanes per cycle)\n", p.gcnArchName, cus);
    all<1>(cus, d_out, d_clk); all<2>(cus, d_out, d_clk); all<3>(cus, d_out, d_clk); all<4>(cus, d_out, d_clk); all<6>(cus, d_out, d_clk);
    return 0;
}
