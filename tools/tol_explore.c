/*
 * tol_explore.c -- CPU emulation of RELAXED evaluations of the three SRCNN layers, to see which of the reference's
 * roundings (src/libsrcnn.cpp:395-410 conv 9x9, :433-437 conv 1x1, :500-517 conv 5x5) can be given up inside a
 * |dY| <= 1e-4 budget.  DEVELOPMENT TOOL (a CPU emulation with the product's weight table; never part of the product, never used as a checker).  Each relaxed form
 * below is the exact arithmetic of a candidate device instruction sequence:
 *   layer 1/2  "fma"  : acc = fmaf(w, x, acc)            (MFMA with C = acc: one rounding per tap instead of two)
 *              masks  : only the taps / channels whose weights are smallest take the fma form, the rest stay strict
 *   layer 3    "x64"  : a = fma((double)w, (double)c, a)  (v_fma_f64 on widened operands: the product is exact instead of
 *                        rounded to fp32; everything else -- per-channel fp64 sum, fp32 running sum -- as the reference)
 *              "f32"  : a = fmaf(w, c, a) in fp32         (v_fma_f32 chain, the existing FAST tier)
 *   "exact": all three layers in fp64 (what ANY reformulation that is merely accurate converges to)
 * Input: a raw float32 plane (the upscaled Y, W x H).  Output: one line per experiment: max|dY| and mean|dY| against the
 * strict result (= the reference, bit for bit), and the fraction of samples that differ at all.
 *
 *   gcc -O2 -fopenmp -mavx2 -mfma -ffp-contract=off tools/tol_explore.c -lm -o /tmp/tol_explore
 *   /tmp/tol_explore plane.f32 W H
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define C1 64
#define C2 32
static const uint32_t k_weight_bits[8129] = {
#include "../libsrcnn_amd/csrc/srcnn_weights.inc"
};
#define OFF_B1 0
#define OFF_W1 (OFF_B1 + 64)
#define OFF_B2 (OFF_W1 + 64 * 81)
#define OFF_W2 (OFF_B2 + 32)
#define OFF_B3 (OFF_W2 + 32 * 64)
#define OFF_W3 (OFF_B3 + 1)
static const float* WT(void) { return (const float*)(const void*)k_weight_bits; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static float w1t[81][C1] __attribute__((aligned(32)));   /* [tap][k]  */
static float w2t[C1][C2] __attribute__((aligned(32)));   /* [f][m]    */

static void prep(void)
{
    const float* w = WT();
    for (int k = 0; k < C1; ++k)
        for (int t = 0; t < 81; ++t) w1t[t][k] = w[OFF_W1 + k * 81 + t];
    for (int m = 0; m < C2; ++m)
        for (int f = 0; f < C1; ++f) w2t[f][m] = w[OFF_W2 + m * 64 + f];
}

/* layers 1+2: tap t of layer 1 is fused iff fma1[t]; channel f of layer 2 iff fma2[f].  c2: [32][H*W] */
static void conv12(const float* Y, int W, int H, const unsigned char* fma1, const unsigned char* fma2, float* c2)
{
    const float* w = WT();
    const size_t N = (size_t)W * H;
#pragma omp parallel for schedule(dynamic, 4)
    for (int r = 0; r < H; ++r) {
        for (int c = 0; c < W; ++c) {
            float acc[C1] __attribute__((aligned(32)));
            for (int k = 0; k < C1; ++k) acc[k] = 0.f;
            for (int t = 0; t < 81; ++t) {
                const float y = Y[(size_t)clampi(r + t / 9 - 4, 0, H - 1) * W + clampi(c + t % 9 - 4, 0, W - 1)];
                const float* wr = w1t[t];
                if (fma1[t]) for (int k = 0; k < C1; ++k) acc[k] = __builtin_fmaf(wr[k], y, acc[k]);
                else         for (int k = 0; k < C1; ++k) acc[k] = acc[k] + wr[k] * y;
            }
            for (int k = 0; k < C1; ++k) { const float v = acc[k] + w[OFF_B1 + k]; acc[k] = v >= 0.f ? v : 0.f; }
            float a2[C2] __attribute__((aligned(32)));
            for (int m = 0; m < C2; ++m) a2[m] = 0.f;
            for (int f = 0; f < C1; ++f) {
                const float x = acc[f];
                const float* wr = w2t[f];
                if (fma2[f]) for (int m = 0; m < C2; ++m) a2[m] = __builtin_fmaf(x, wr[m], a2[m]);
                else         for (int m = 0; m < C2; ++m) a2[m] = a2[m] + x * wr[m];
            }
            for (int m = 0; m < C2; ++m) { const float v = a2[m] + w[OFF_B2 + m]; c2[(size_t)m * N + (size_t)r * W + c] = v >= 0.f ? v : 0.f; }
        }
    }
}

/* layer 3.  mode3[ch]: 0 strict, 1 exact product in fp64 (x64), 2 fp32 fma chain */
static size_t g_mag[4];     /* samples whose largest |fp32 running sum| in layer 3 reaches 256 / 512 / 1024 / 2048 */
static void conv3(const float* c2, int W, int H, const unsigned char* mode3, float* out)
{
    size_t mag0 = 0, mag1 = 0, mag2 = 0, mag3 = 0;
    const float* w = WT();
    const size_t N = (size_t)W * H;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : mag0, mag1, mag2, mag3)
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            float sum = 0.f, big = 0.f;
            for (int i = 0; i < C2; ++i) {
                const float* pl = c2 + (size_t)i * N;
                const float* k = w + OFF_W3 + i * 25;
                double a = 0.0;
                float af = 0.f;
                for (int y = 0; y < 5; ++y)
                    for (int x = 0; x < 5; ++x) {
                        const float v = pl[(size_t)clampi(r + y - 2, 0, H - 1) * W + clampi(c + x - 2, 0, W - 1)];
                        const float kw = k[x * 5 + y];
                        if (mode3[i] == 0) { const float p = kw * v; a = a + (double)p; }
                        else if (mode3[i] == 1) a = fma((double)kw, (double)v, a);
                        else af = __builtin_fmaf(kw, v, af);
                    }
                if (mode3[i] == 2) a = (double)af;
                sum = (float)((double)sum + a);
                big = fabsf(sum) > big ? fabsf(sum) : big;
            }
            mag0 += big >= 256.f; mag1 += big >= 512.f; mag2 += big >= 1024.f; mag3 += big >= 2048.f;
            float t = sum + w[OFF_B3];
            t = t > 0.f ? t : 0.f;
            t = t < 255.f ? t : 255.f;
            out[(size_t)r * W + c] = t;
        }
    g_mag[0] = mag0; g_mag[1] = mag1; g_mag[2] = mag2; g_mag[3] = mag3;
}

/* everything in fp64, rounded once at the end */
static void exact_path(const float* Y, int W, int H, float* out)
{
    const float* w = WT();
    const size_t N = (size_t)W * H;
    double* c2 = malloc(sizeof(double) * N * C2);
#pragma omp parallel for schedule(dynamic, 4)
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            double acc[C1];
            for (int k = 0; k < C1; ++k) acc[k] = 0.0;
            for (int t = 0; t < 81; ++t) {
                const double y = Y[(size_t)clampi(r + t / 9 - 4, 0, H - 1) * W + clampi(c + t % 9 - 4, 0, W - 1)];
                for (int k = 0; k < C1; ++k) acc[k] += (double)w1t[t][k] * y;
            }
            for (int k = 0; k < C1; ++k) { acc[k] += w[OFF_B1 + k]; if (acc[k] < 0) acc[k] = 0; }
            for (int m = 0; m < C2; ++m) {
                double a = 0;
                for (int f = 0; f < C1; ++f) a += acc[f] * (double)w2t[f][m];
                a += w[OFF_B2 + m];
                c2[(size_t)m * N + (size_t)r * W + c] = a > 0 ? a : 0;
            }
        }
#pragma omp parallel for schedule(dynamic, 4)
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            double s = 0;
            for (int i = 0; i < C2; ++i)
                for (int y = 0; y < 5; ++y)
                    for (int x = 0; x < 5; ++x)
                        s += (double)w[OFF_W3 + i * 25 + x * 5 + y] * c2[(size_t)i * N + (size_t)clampi(r + y - 2, 0, H - 1) * W + clampi(c + x - 2, 0, W - 1)];
            s += w[OFF_B3];
            s = s > 0 ? s : 0; s = s < 255 ? s : 255;
            out[(size_t)r * W + c] = (float)s;
        }
    free(c2);
}

static void report(const char* name, const float* ref, const float* got, size_t N)
{
    double mx = 0, sm = 0; size_t nd = 0;
    for (size_t i = 0; i < N; ++i) {
        const double d = fabs((double)ref[i] - (double)got[i]);
        if (d > mx) mx = d;
        sm += d;
        nd += d != 0;
    }
    printf("%-44s max %.3e  mean %.3e  differ %.4f\n", name, mx, sm / N, (double)nd / N);
    fflush(stdout);
}

static int cmp_idx(const void* a, const void* b, void* key)
{
    const double* k = key; const int ia = *(const int*)a, ib = *(const int*)b;
    return k[ia] < k[ib] ? -1 : (k[ia] > k[ib] ? 1 : 0);
}

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s plane.f32 W H [quick]\n", argv[0]); return 2; }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    const int quick = argc > 4;
    const size_t N = (size_t)W * H;
    float* Y = malloc(4 * N);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(Y, 4, N, f) != N) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    prep();
    const float* w = WT();

    /* rank layer-1 taps, layer-2 input channels and layer-3 channels by the size of their weights */
    double k1[81], k2[C1], k3[C2]; int o1[81], o2[C1], o3[C2];
    for (int t = 0; t < 81; ++t) { k1[t] = 0; o1[t] = t; for (int k = 0; k < C1; ++k) k1[t] = fmax(k1[t], fabs(w1t[t][k])); }
    for (int ff = 0; ff < C1; ++ff) { k2[ff] = 0; o2[ff] = ff; for (int m = 0; m < C2; ++m) k2[ff] = fmax(k2[ff], fabs(w2t[ff][m])); }
    for (int i = 0; i < C2; ++i) { k3[i] = 0; o3[i] = i; for (int t = 0; t < 25; ++t) k3[i] = fmax(k3[i], fabs(w[OFF_W3 + i * 25 + t])); }
    qsort_r(o1, 81, sizeof(int), cmp_idx, k1); qsort_r(o2, C1, sizeof(int), cmp_idx, k2); qsort_r(o3, C2, sizeof(int), cmp_idx, k3);
    printf("# layer-1 taps by max|w| over channels: smallest %.4g, median %.4g, largest %.4g\n", k1[o1[0]], k1[o1[40]], k1[o1[80]]);
    printf("# layer-2 input channels by max|w|:     smallest %.4g, median %.4g, largest %.4g\n", k2[o2[0]], k2[o2[32]], k2[o2[63]]);
    printf("# layer-3 channels by max|w|:           smallest %.4g, median %.4g, largest %.4g\n", k3[o3[0]], k3[o3[16]], k3[o3[31]]);

    float* c2s = malloc(4 * N * C2); float* c2v = malloc(4 * N * C2);
    float* ref = malloc(4 * N); float* out = malloc(4 * N);
    unsigned char z1[81] = {0}, z2[C1] = {0}, z3[C2] = {0}, m1[81], m2[C1], m3[C2];
    conv12(Y, W, H, z1, z2, c2s);
    conv3(c2s, W, H, z3, ref);
    printf("# layer 3 (strict): fraction of samples whose fp32 running sum reaches |s| >= 256: %.4f, >= 512: %.4f, >= 1024: %.5f, >= 2048: %.6f\n"
           "#   (one ulp there is 3.1e-5 / 6.1e-5 / 1.2e-4 / 2.4e-4: a single rounding that falls the other way costs that much)\n",
           (double)g_mag[0] / N, (double)g_mag[1] / N, (double)g_mag[2] / N, (double)g_mag[3] / N);

    exact_path(Y, W, H, out);                    report("exact (fp64 everywhere) vs reference", ref, out, N);

    /* layer 3 alone */
    memset(m3, 1, C2); conv3(c2s, W, H, m3, out); report("L3 x64 (exact products)", ref, out, N);
    memset(m3, 2, C2); conv3(c2s, W, H, m3, out); report("L3 f32 fma chain", ref, out, N);
    for (int n = 8; n <= 24 && !quick; n += 8) {
        memset(m3, 0, C2); for (int i = 0; i < n; ++i) m3[o3[i]] = 1;
        conv3(c2s, W, H, m3, out);
        char nm[64]; snprintf(nm, sizeof nm, "L3 x64 on the %d smallest-weight channels", n); report(nm, ref, out, N);
    }
    /* layer 1 */
    memset(m1, 1, 81); conv12(Y, W, H, m1, z2, c2v); conv3(c2v, W, H, z3, out); report("L1 fma (all 81 taps)", ref, out, N);
    memset(m3, 1, C2); conv3(c2v, W, H, m3, out); report("L1 fma + L3 x64", ref, out, N);
    for (int n = 20; n <= 60 && !quick; n += 20) {
        memset(m1, 0, 81); for (int i = 0; i < n; ++i) m1[o1[i]] = 1;
        conv12(Y, W, H, m1, z2, c2v); conv3(c2v, W, H, z3, out);
        char nm[64]; snprintf(nm, sizeof nm, "L1 fma on the %d smallest-weight taps", n); report(nm, ref, out, N);
    }
    /* layer 2 */
    memset(m2, 1, C1); conv12(Y, W, H, z1, m2, c2v); conv3(c2v, W, H, z3, out); report("L2 fma (all 64 channels)", ref, out, N);
    memset(m3, 1, C2); conv3(c2v, W, H, m3, out); report("L2 fma + L3 x64", ref, out, N);
    for (int n = 16; n <= 48 && !quick; n += 16) {
        memset(m2, 0, C1); for (int i = 0; i < n; ++i) m2[o2[i]] = 1;
        conv12(Y, W, H, z1, m2, c2v); conv3(c2v, W, H, z3, out);
        char nm[64]; snprintf(nm, sizeof nm, "L2 fma on the %d smallest-weight channels", n); report(nm, ref, out, N);
    }
    /* all */
    memset(m1, 1, 81); memset(m2, 1, C1); conv12(Y, W, H, m1, m2, c2v);
    conv3(c2v, W, H, z3, out); report("L1 fma + L2 fma", ref, out, N);
    memset(m3, 1, C2); conv3(c2v, W, H, m3, out); report("L1 fma + L2 fma + L3 x64", ref, out, N);
    memset(m3, 2, C2); conv3(c2v, W, H, m3, out); report("L1 fma + L2 fma + L3 f32 (= SRCNN_MODE_FAST)", ref, out, N);
    return 0;
}
