// vgpr_banks.hip -- does v_pk_add_f32 pay for VGPR bank conflicts on gfx950?
// 16 packed adds per iteration, dst/src0 = v[32+2i:33+2i], src1 = v[96+OFF+2i : ...]; OFF = 0 puts both
// 64-bit operands on the same bank pair (register index mod 4), OFF = 2 on the other pair.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>

#define ADD(i, off) "v_pk_add_f32 v[" #i ":" #i "+1], v[" #i ":" #i "+1], v[" #i "+64+" #off ":" #i "+65+" #off "]\n\t"
#define ADDS(off) ADD(32, off) ADD(34, off) ADD(36, off) ADD(38, off) ADD(40, off) ADD(42, off) ADD(44, off) ADD(46, off) \
                  ADD(48, off) ADD(50, off) ADD(52, off) ADD(54, off) ADD(56, off) ADD(58, off) ADD(60, off) ADD(62, off)
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
             "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129"

template <int OFF>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* clk, int iters)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (OFF == 0) asm volatile(ADDS(0) ::: CLOB);
        else asm volatile(ADDS(2) ::: CLOB);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float v; asm volatile("v_mov_b32 %0, v32" : "=v"(v));
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int OFF>
void run(int bpc, int cus, float* d_out, unsigned long long* d_clk)
{
    const int iters = 100000, grid = cus * bpc;
    hipLaunchKernelGGL(k<OFF>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, 100);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OFF>, dim3(grid), dim3(256), 0, 0, d_out, d_clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(grid);
    hipMemcpy(c.data(), d_clk, grid * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    printf("src1 offset %d regs: waves/SIMD=%d  %.2f ms  shader cycles per 16 pk_add per wave = %.1f  => %.2f cycles per pk_add per SIMD\n",
           OFF, bpc, ms, (double)c[grid / 2] / iters, (double)c[grid / 2] / iters / 16.0 / bpc);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d_out; hipMalloc(&d_out, sizeof(float) * 256 * cus * 4);
    unsigned long long* d_clk; hipMalloc(&d_clk, 8 * cus * 4);
    for (int bpc : {1, 2, 3}) { run<0>(bpc, cus, d_out, d_clk); run<2>(bpc, cus, d_out, d_clk); }
    return 0;
}
