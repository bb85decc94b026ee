/*
 * ref_shim.cpp -- thin extern "C" doorway into the REAL reference, for pinning the oracle.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as srcnn_oracle.c).  Built only where /root/reference
 * exists (the dev container) by oracle/Makefile into oracle/_ref/libsrcnn_ref.so, together with
 * the reference's own frawscale.cpp / libsrcnn.cpp compiled from where they lie with the flags of
 * Makefiles/Makefile.linux:26-35 (-fPIC -DBUILDING_DLL -fopenmp -O2).  Nothing from the reference
 * tree is copied into this repo; this file only declares the reference's exported entry points
 * (libsrcnn.cpp:95-104 prototypes, the types at :61-87) and forwards to them.
 */
#include <cstring>
#include "libsrcnn.h"   /* -I/root/reference/src */
#include "frawscale.h"
#include "convdata.h"

namespace libsrcnn {
/* mirror of the plane descriptor (libsrcnn.cpp:69-75); layout must match for the calls below */
typedef struct { unsigned width; unsigned height; unsigned depth; float* buff; } ImgF32;
typedef ImgF32 ImgConv1Layers[CONV1_FILTERS];
typedef ImgF32 ImgConv2Layers[CONV2_FILTERS];
void convolution99(ImgF32& src, ImgF32& dst, const KernelMat99 kernel, float bias);
void convolution11(ImgConv1Layers& src, ImgF32& dst, const ConvKernel1 kernel, float bias);
void convolution55(ImgConv2Layers& src, ImgF32& dst, const ConvKernel32_55 kernel, float bias);
}

static FRAWGenericFilter* make_filter(int kind)
{
    switch (kind) {
    case 0: return new FRAWBoxFilter;
    case 1: return new FRAWBilinearFilter;
    case 3: return new FRAWLanczos3Filter;
    case 4: return new FRAWBSplineFilter;
    default: return new FRAWBicubicFilter;
    }
}

extern "C" {

int ref_resample(const float* src, unsigned sw, unsigned sh, unsigned dw, unsigned dh, float* dst, int filter)
{
    FRAWGenericFilter* f = make_filter(filter);
    FRAWResizeEngine eng(f);
    float* out = NULL;
    unsigned n = eng.scale(src, sw, sh, dw, dh, &out);
    if (out) { memcpy(dst, out, sizeof(float) * (size_t)dw * dh); delete[] out; }
    delete f;
    return n ? 0 : -1;
}

void ref_conv1(const float* y, unsigned w, unsigned h, float* out)
{
    libsrcnn::ImgF32 src = { w, h, 1, const_cast<float*>(y) };
    #pragma omp parallel for
    for (int k = 0; k < CONV1_FILTERS; ++k) {
        libsrcnn::ImgF32 dst = { w, h, 1, out + (size_t)k * w * h };
        libsrcnn::convolution99(src, dst, weights_conv1_data[k], biases_conv1[k]);
    }
}

void ref_conv2(const float* in, unsigned w, unsigned h, float* out)
{
    libsrcnn::ImgConv1Layers src;
    for (int k = 0; k < CONV1_FILTERS; ++k) { libsrcnn::ImgF32 p = { w, h, 1, const_cast<float*>(in) + (size_t)k * w * h }; src[k] = p; }
    #pragma omp parallel for
    for (int m = 0; m < CONV2_FILTERS; ++m) {
        libsrcnn::ImgF32 dst = { w, h, 1, out + (size_t)m * w * h };
        libsrcnn::convolution11(src, dst, weights_conv2_data[m], biases_conv2[m]);
    }
}

void ref_conv3(const float* in, unsigned w, unsigned h, float* out)
{
    libsrcnn::ImgConv2Layers src;
    for (int m = 0; m < CONV2_FILTERS; ++m) { libsrcnn::ImgF32 p = { w, h, 1, const_cast<float*>(in) + (size_t)m * w * h }; src[m] = p; }
    libsrcnn::ImgF32 dst = { w, h, 1, out };
    libsrcnn::convolution55(src, dst, weights_conv3_data, biases_conv3);
}

/* bicubic(filter) -> conv1 -> conv2 -> conv3, exactly the sequence of libsrcnn.cpp:716-723,785-846 */
int ref_y_path(const float* y, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter,
               float* out, float* up_opt, float* c1_opt, float* c2_opt)
{
    const size_t n = (size_t)dw * dh;
    float* up = up_opt ? up_opt : new float[n];
    float* c1 = c1_opt ? c1_opt : new float[n * CONV1_FILTERS];
    float* c2 = c2_opt ? c2_opt : new float[n * CONV2_FILTERS];
    int rc = ref_resample(y, w, h, dw, dh, up, filter);
    if (rc == 0) { ref_conv1(up, dw, dh, c1); ref_conv2(c1, dw, dh, c2); ref_conv3(c2, dw, dh, out); }
    if (!up_opt) delete[] up;
    if (!c1_opt) delete[] c1;
    if (!c2_opt) delete[] c2;
    return rc;
}

int ref_y_upscale2x(const float* y, unsigned w, unsigned h, float* out)
{
    return ref_y_path(y, w, h, 2 * w, 2 * h, 2, out, NULL, NULL, NULL);
}

/* the public API itself: ConfigureFilterSRCNN + ProcessSRCNN (libsrcnn.h:46-54) */
int ref_process(const unsigned char* rgb, unsigned w, unsigned h, unsigned d, float mul, int filter, int step,
                unsigned char* out, unsigned out_cap, unsigned* out_sz,
                unsigned char* conv, unsigned conv_cap, unsigned* conv_sz)
{
    ConfigureFilterSRCNN((SRCNNFilterType)filter, step != 0);
    unsigned char* ob = NULL; unsigned osz = 0;
    unsigned char* cb = NULL; unsigned csz = 0;
    int rc = ProcessSRCNN(rgb, w, h, d, mul, ob, osz, conv ? &cb : NULL, conv ? &csz : NULL);
    if (rc == 0) {
        if (out_sz) *out_sz = osz;
        if (ob && osz <= out_cap) memcpy(out, ob, osz);
        if (conv_sz) *conv_sz = csz;
        if (cb && csz <= conv_cap) memcpy(conv, cb, csz);
    }
    delete[] ob; delete[] cb;
    return rc;
}

void ref_get_weights(float* out)
{
    float* p = out;
    memcpy(p, biases_conv1, sizeof biases_conv1); p += 64;
    memcpy(p, weights_conv1_data, sizeof weights_conv1_data); p += 64 * 81;
    memcpy(p, biases_conv2, sizeof biases_conv2); p += 32;
    memcpy(p, weights_conv2_data, sizeof weights_conv2_data); p += 32 * 64;
    *p++ = biases_conv3;
    memcpy(p, weights_conv3_data, sizeof weights_conv3_data);
}

} /* extern "C" */
