/*
 * srcnn_oracle.c -- CPU restatement of the rageworx/libsrcnn Y-channel path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path in
 * libsrcnn_amd/csrc.  It may be imported / linked / executed only from tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py.  The product never calls it
 * and has no CPU fallback.
 *
 * Pinning: every function below is checked bit-for-bit against the reference itself
 * (oracle/_ref, compiled from /root/reference/src by oracle/Makefile) in
 * tests/test_oracle_vs_ref.py, and against the committed golden fixtures in tests/golden/
 * (among them the reference's own Pictures/butterfly.png -> butterfly_srcnn.png /
 * butterfly_srcnn_convolution.png pair) in tests/test_oracle_golden.py.
 *
 * Arithmetic contract (what "bit-exact" means here; all cites are /root/reference/src):
 *   - conv 9x9 (libsrcnn.cpp:350-422): fp32 product, fp32 add, taps in row-major order
 *     starting from 0.0f, bias added last, ReLU.  Borders: clamp-to-edge of the INPUT plane.
 *   - conv 1x1 (libsrcnn.cpp:424-447): fp32 product/add over channels 0..63, bias last, ReLU.
 *   - conv 5x5 (libsrcnn.cpp:449-529): per channel an fp64 accumulator of fp32 products
 *     (taps ordered row-major over the window, weight index [chan][col][row]), folded into an
 *     fp32 running sum via (float)((double)sum + acc); bias; clamp to [0,255].
 *     Borders: clamp-to-edge of the conv-2 ACTIVATIONS.
 *   - resampler (frawscale.cpp:8-112, 162-385): fp64 weights, fp64 multiply then fp64 add
 *     (no FMA), fp32 store after each separable pass, truncate-and-renormalise borders.
 * Build with -ffp-contract=off and without -march flags that enable FMA.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define C1 64 /* conv-1 output channels  (convdata.h:5)  */
#define C2 32 /* conv-2 output channels  (convdata.h:8)  */

static const uint32_t k_weight_bits[8129] = {
#include "oracle_weights.inc"
};

#define OFF_B1 0
#define OFF_W1 (OFF_B1 + 64)
#define OFF_B2 (OFF_W1 + 64 * 81)
#define OFF_W2 (OFF_B2 + 32)
#define OFF_B3 (OFF_W2 + 32 * 64)
#define OFF_W3 (OFF_B3 + 1)

static inline const float* wtab(void) { return (const float*)(const void*)k_weight_bits; }

/* Copy of the 8129 weights in blob order, for tests that want to feed them elsewhere. */
void oracle_get_weights(float* out) { memcpy(out, k_weight_bits, sizeof k_weight_bits); }

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------------------------
 * Resampler.  filter ids follow SRCNNFilterType (libsrcnn.h:37-44):
 *   0 box(nearest) 1 bilinear 2 bicubic(Mitchell B=C=1/3) 3 lanczos3 4 b-spline
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int kind;
    double width;
    double p0, p2, p3, q0, q1, q2, q3; /* Mitchell polynomial pieces (frawscale.h:97-106) */
} filt_t;

static void filt_init(filt_t* f, int kind)
{
    memset(f, 0, sizeof *f);
    f->kind = kind;
    switch (kind) {
    case 0: f->width = 0.5; break;  /* frawscale.h:64 */
    case 1: f->width = 1.0; break;  /* frawscale.h:75 */
    case 3: f->width = 3.0; break;  /* frawscale.h:127 */
    case 4: f->width = 2.0; break;  /* frawscale.h:157 */
    default: {                      /* frawscale.h:95-107 */
        const double b = 1 / (double)3, c = 1 / (double)3;
        f->kind = 2;
        f->width = 2.0;
        f->p0 = (6 - 2 * b) / 6;
        f->p2 = (-18 + 12 * b + 6 * c) / 6;
        f->p3 = (12 - 9 * b - 6 * c) / 6;
        f->q0 = (8 * b + 24 * c) / 6;
        f->q1 = (-12 * b - 48 * c) / 6;
        f->q2 = (6 * b + 30 * c) / 6;
        f->q3 = (-b - 6 * c) / 6;
    } break;
    }
}

static double sinc_pi(double v)
{
    if (v != 0) {
        v *= 3.1415926535897932384626433832795;
        return sin(v) / v;
    }
    return 1;
}

static double filt_eval(const filt_t* f, double t)
{
    t = fabs(t);
    switch (f->kind) {
    case 0: return t <= f->width ? 1.0 : 0.0;                  /* frawscale.h:68 */
    case 1: return t < f->width ? f->width - t : 0.0;          /* frawscale.h:79-82 */
    case 3: return t < f->width ? sinc_pi(t) * sinc_pi(t / f->width) : 0.0; /* :131-138 */
    case 4:                                                     /* frawscale.h:161-170 */
        if (t < 1) return (4 + t * t * (-6 + 3 * t)) / 6;
        if (t < 2) { double u = 2 - t; return u * u * u / 6; }
        return 0;
    default:                                                    /* frawscale.h:111-120 */
        if (t < 1) return f->p0 + t * t * (f->p2 + t * f->p3);
        if (t < 2) return f->q0 + t * (f->q1 + t * (f->q2 + t * f->q3));
        return 0;
    }
}

/*
 * Contribution table for one axis (frawscale.cpp:8-112).
 * left/right: inclusive source range per destination coordinate; w: window*dst doubles,
 * row u holds weights for taps left[u]..  Returns the window size (row stride of w).
 */
int oracle_axis_window(int filter, unsigned dst_len, unsigned src_len)
{
    filt_t f;
    filt_init(&f, filter);
    const double scale = (double)dst_len / (double)src_len;
    const double span = scale < 1.0 ? f.width / scale : f.width;
    return 2 * (int)ceil(span) + 1;
}

void oracle_axis_table(int filter, unsigned dst_len, unsigned src_len,
                       int* left, int* right, double* w /* dst_len x (window+1) */)
{
    filt_t f;
    filt_init(&f, filter);
    const double scale = (double)dst_len / (double)src_len;
    double span, fscale = 1.0;
    if (scale < 1.0) { span = f.width / scale; fscale = scale; }
    else span = f.width;
    const int window = 2 * (int)ceil(span) + 1;
    const int stride = window + 1;
    const double shift = (0.5 / scale) - 0.5;

    for (unsigned u = 0; u < dst_len; ++u) {
        double* row = w + (size_t)u * stride;
        const double center = (double)u / scale + shift;
        int lo = (int)floor(center - span);
        if (lo < 0) lo = 0;
        int hi = (int)ceil(center + span);
        if (hi > (int)src_len - 1) hi = (int)src_len - 1;
        if (hi - lo + 1 > window) {
            /* frawscale.cpp:57 -- "uSrcSize - 1 / 2" is uSrcSize - 0 */
            if (lo < (int)src_len - 1 / 2) lo++;
            else hi--;
        }
        left[u] = lo;
        right[u] = hi;
        double total = 0;
        for (int s = lo; s <= hi; ++s) {
            const double wt = fscale * filt_eval(&f, fscale * (center - (double)s));
            row[s - lo] = wt;
            total += wt;
        }
        if (total > 0 && total != 1) {
            for (int s = lo; s <= hi; ++s) row[s - lo] /= total;
            int t = hi - lo;
            while (row[t] == 0) { /* drop trailing zero taps only (frawscale.cpp:95-107) */
                right[u]--;
                t--;
                if (right[u] == left[u]) break;
            }
        }
    }
}

static void pass_rows(const float* src, unsigned rows, unsigned src_w, float* dst, unsigned dst_w,
                      int filter)
{   /* horizontal pass, frawscale.cpp:288-332 */
    const int stride = oracle_axis_window(filter, dst_w, src_w) + 1;
    int* lo = (int*)malloc(sizeof(int) * dst_w);
    int* hi = (int*)malloc(sizeof(int) * dst_w);
    double* wt = (double*)malloc(sizeof(double) * (size_t)dst_w * stride);
    oracle_axis_table(filter, dst_w, src_w, lo, hi, wt);
#pragma omp parallel for
    for (long y = 0; y < (long)rows; ++y) {
        const float* in = src + (size_t)y * src_w;
        float* out = dst + (size_t)y * dst_w;
        for (unsigned x = 0; x < dst_w; ++x) {
            const double* wr = wt + (size_t)x * stride;
            double acc = 0.0;
            for (int s = lo[x]; s <= hi[x]; ++s) {
                const double px = in[s];
                acc += wr[s - lo[x]] * px;
            }
            out[x] = (float)acc;
        }
    }
    free(lo); free(hi); free(wt);
}

static void pass_cols(const float* src, unsigned w, unsigned src_h, float* dst, unsigned dst_h,
                      int filter)
{   /* vertical pass, frawscale.cpp:335-385 */
    const int stride = oracle_axis_window(filter, dst_h, src_h) + 1;
    int* lo = (int*)malloc(sizeof(int) * dst_h);
    int* hi = (int*)malloc(sizeof(int) * dst_h);
    double* wt = (double*)malloc(sizeof(double) * (size_t)dst_h * stride);
    oracle_axis_table(filter, dst_h, src_h, lo, hi, wt);
#pragma omp parallel for
    for (long y = 0; y < (long)dst_h; ++y) {
        const double* wr = wt + (size_t)y * stride;
        float* out = dst + (size_t)y * w;
        for (unsigned x = 0; x < w; ++x) {
            double acc = 0.0;
            for (int s = lo[y]; s <= hi[y]; ++s) {
                const double px = src[(size_t)s * w + x];
                acc += wr[s - lo[y]] * px;
            }
            out[x] = (float)acc;
        }
    }
    free(lo); free(hi); free(wt);
}

/*
 * FRAWResizeEngine::scale (frawscale.cpp:162-286).  Upscale in x => vertical pass first.
 * The identity-size branch of the reference copies only sizeof(unsigned short) bytes per pixel
 * (frawscale.cpp:185-193); that quirk is reproduced by leaving the rest of dst as it was
 * handed in (callers pass a zeroed buffer), see tests.
 */
int oracle_resample(const float* src, unsigned sw, unsigned sh, unsigned dw, unsigned dh,
                    float* dst, int filter)
{
    if (!src || !dst || !sw || !sh || !dw || !dh) return -1;
    if (sw == dw && sh == dh) {
        memcpy(dst, src, (size_t)sw * sh * sizeof(unsigned short));
        return 0;
    }
    if (dw <= sw) { /* horizontal first (frawscale.cpp:195-237) */
        const float* mid = src;
        float* tmp = NULL;
        if (sw != dw) {
            tmp = (sh != dh) ? (float*)malloc(sizeof(float) * (size_t)dw * sh) : dst;
            pass_rows(src, sh, sw, tmp, dw, filter);
            mid = tmp;
        }
        if (sh != dh) pass_cols(mid, dw, sh, dst, dh, filter);
        if (tmp && tmp != dst) free(tmp);
    } else {        /* vertical first (frawscale.cpp:238-278) */
        const float* mid = src;
        float* tmp = NULL;
        if (sh != dh) {
            tmp = (float*)malloc(sizeof(float) * (size_t)sw * dh);
            pass_cols(src, sw, sh, tmp, dh, filter);
            mid = tmp;
        }
        pass_rows(mid, dh, sw, dst, dw, filter);
        if (tmp) free(tmp);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * The three convolutions.
 * ---------------------------------------------------------------------------------------- */

/* libsrcnn.cpp:350-422 for all 64 filters (driver loop :791-798).  out: 64 planes of w*h. */
void oracle_conv1(const float* y, unsigned w, unsigned h, float* out)
{
    const float* W = wtab() + OFF_W1;
    const float* B = wtab() + OFF_B1;
    const unsigned pw = w + 8, ph = h + 8;
    float* pad = (float*)malloc(sizeof(float) * (size_t)pw * ph);
    for (unsigned r = 0; r < ph; ++r)
        for (unsigned c = 0; c < pw; ++c)
            pad[(size_t)r * pw + c] =
                y[(size_t)clampi((int)r - 4, 0, (int)h - 1) * w + clampi((int)c - 4, 0, (int)w - 1)];
#pragma omp parallel for
    for (int k = 0; k < C1; ++k) {
        const float* ker = W + k * 81;
        float* dst = out + (size_t)k * w * h;
        for (unsigned r = 0; r < h; ++r)
            for (unsigned c = 0; c < w; ++c) {
                float acc = 0;
                for (int i = 0; i < 9; ++i)
                    for (int j = 0; j < 9; ++j)
                        acc += ker[i * 9 + j] * pad[(size_t)(r + i) * pw + (c + j)];
                acc += B[k];
                dst[(size_t)r * w + c] = (acc >= 0) ? acc : 0;
            }
    }
    free(pad);
}

/* libsrcnn.cpp:424-447 for all 32 outputs (driver loop :817-824). in: 64 planes, out: 32. */
void oracle_conv2(const float* in, unsigned w, unsigned h, float* out)
{
    const float* W = wtab() + OFF_W2;
    const float* B = wtab() + OFF_B2;
    const size_t n = (size_t)w * h;
#pragma omp parallel for
    for (int m = 0; m < C2; ++m) {
        const float* ker = W + m * C1;
        float* dst = out + (size_t)m * n;
        for (size_t p = 0; p < n; ++p) {
            float acc = 0;
            for (int f = 0; f < C1; ++f) acc += in[(size_t)f * n + p] * ker[f];
            acc += B[m];
            dst[p] = (acc >= 0) ? acc : 0;
        }
    }
}

/* libsrcnn.cpp:449-529. in: 32 planes, out: 1 plane. */
void oracle_conv3(const float* in, unsigned w, unsigned h, float* out)
{
    const float* W = wtab() + OFF_W3;
    const float bias = wtab()[OFF_B3];
    const unsigned pw = w + 4, ph = h + 4;
    const size_t n = (size_t)w * h, pn = (size_t)pw * ph;
    float* pad = (float*)malloc(sizeof(float) * pn * C2);
#pragma omp parallel for
    for (int m = 0; m < C2; ++m)
        for (unsigned r = 0; r < ph; ++r)
            for (unsigned c = 0; c < pw; ++c)
                pad[m * pn + (size_t)r * pw + c] =
                    in[m * n + (size_t)clampi((int)r - 2, 0, (int)h - 1) * w +
                       clampi((int)c - 2, 0, (int)w - 1)];
#pragma omp parallel for
    for (long r = 0; r < (long)h; ++r)
        for (unsigned c = 0; c < w; ++c) {
            float sum = 0;
            for (int m = 0; m < C2; ++m) {
                double acc = 0;
                for (int dy = 0; dy < 5; ++dy)
                    for (int dx = 0; dx < 5; ++dx) /* weight index is [chan][col][row] (:512) */
                        acc += W[m * 25 + dx * 5 + dy] * pad[m * pn + (size_t)(r + dy) * pw + (c + dx)];
                sum += acc; /* (float)((double)sum + acc) */
            }
            sum += bias;
            sum = sum > 0.f ? sum : 0.f;
            sum = sum < 255.f ? sum : 255.f;
            out[(size_t)r * w + c] = sum;
        }
    free(pad);
}

/*
 * Whole Y path (libsrcnn.cpp:716-723 for plane 0, then :785-846): resample to (dw,dh) with
 * `filter`, three convolutions.  Optional taps of the intermediates for layer-level tests.
 */
int oracle_y_path(const float* y, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter,
                  float* out, float* up_opt, float* c1_opt, float* c2_opt)
{
    const size_t n = (size_t)dw * dh;
    float* up = up_opt ? up_opt : (float*)calloc(n, sizeof(float));
    float* c1 = c1_opt ? c1_opt : (float*)malloc(sizeof(float) * n * C1);
    float* c2 = c2_opt ? c2_opt : (float*)malloc(sizeof(float) * n * C2);
    int rc = -1;
    if (up && c1 && c2) {
        rc = oracle_resample(y, w, h, dw, dh, up, filter);
        if (rc == 0) {
            oracle_conv1(up, dw, dh, c1);
            oracle_conv2(c1, dw, dh, c2);
            oracle_conv3(c2, dw, dh, out);
        }
    }
    if (!up_opt) free(up);
    if (!c1_opt) free(c1);
    if (!c2_opt) free(c2);
    return rc;
}

/* convenience: 2x Mitchell, the path BASELINE.json names */
int oracle_y_upscale2x(const float* y, unsigned w, unsigned h, float* out)
{
    return oracle_y_path(y, w, h, 2 * w, 2 * h, 2, out, NULL, NULL, NULL);
}

/* ------------------------------------------------------------------------------------------
 * Colour shell around the Y path, one doSRCNN pass (libsrcnn.cpp:628-923).
 *   rgb: interleaved u8, d = 3 or 4.  out: (w*m)x(h*m)xd u8.  conv_opt: truncated Y (may be NULL)
 * ---------------------------------------------------------------------------------------- */
int oracle_dosrcnn(const unsigned char* rgb, unsigned w, unsigned h, unsigned d, float mul,
                   int filter, unsigned char* out, unsigned char* conv_opt)
{
    if (d < 3 || d > 4) return -1;
    const size_t n = (size_t)w * h;
    const unsigned dw = (unsigned)((float)w * mul), dh = (unsigned)((float)h * mul);
    const size_t dn = (size_t)dw * dh;
    float* pl[4] = {0, 0, 0, 0};
    float* rs[4] = {0, 0, 0, 0};
    for (unsigned k = 0; k < d; ++k) {
        pl[k] = (float*)malloc(sizeof(float) * n);
        rs[k] = (float*)calloc(dn, sizeof(float));
    }
    for (size_t p = 0; p < n; ++p) { /* libsrcnn.cpp:242-271 */
        const float r = (float)rgb[p * d + 0], g = (float)rgb[p * d + 1], b = (float)rgb[p * d + 2];
        pl[0][p] = (0.299f * r) + (0.587f * g) + (0.114f * b);
        pl[1][p] = 128.f - (0.1687f * r) - (0.3313f * g) + (0.5f * b);
        pl[2][p] = 128.f + (0.5f * r) - (0.4187f * g) - (0.0813f * b);
        if (d == 4) pl[3][p] = (float)rgb[p * d + 3];
    }
    /* chroma/alpha: box when nearest is configured, else bilinear (libsrcnn.cpp:701-713) */
    const int cfilter = (filter == 0) ? 0 : 1;
    for (unsigned k = 1; k < d; ++k) oracle_resample(pl[k], w, h, dw, dh, rs[k], cfilter);
    int rc = oracle_y_path(pl[0], w, h, dw, dh, filter, rs[0], NULL, NULL, NULL);
    if (rc == 0) {
        for (size_t p = 0; p < dn; ++p) { /* libsrcnn.cpp:287-307 */
            const float fy = rs[0][p], cb = rs[1][p] - 128.f, cr = rs[2][p] - 128.f;
            float R = fy + 45.f * cr / 32.f;
            float G = fy - (11.f * cb + 23.f * cr) / 32.f;
            float B = fy + 113.f * cb / 64.f;
            R = 255.f < R ? 255.f : R; G = 255.f < G ? 255.f : G; B = 255.f < B ? 255.f : B;
            out[p * d + 0] = (unsigned char)(0.f > R ? 0.f : R);
            out[p * d + 1] = (unsigned char)(0.f > G ? 0.f : G);
            out[p * d + 2] = (unsigned char)(0.f > B ? 0.f : B);
            if (d == 4) {
                float A = 255.f < rs[3][p] ? 255.f : rs[3][p];
                out[p * d + 3] = (unsigned char)(0.f > A ? 0.f : A);
            }
            if (conv_opt) conv_opt[p] = (unsigned char)rs[0][p]; /* libsrcnn.cpp:897-901 */
        }
    }
    for (unsigned k = 0; k < d; ++k) { free(pl[k]); free(rs[k]); }
    return rc;
}
