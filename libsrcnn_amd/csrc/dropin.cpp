// dropin.cpp -- ConfigureFilterSRCNN / ProcessSRCNN with the reference's C++ linkage, on top of the
// C ABI.  Mirrors the control flow of src/libsrcnn.cpp:930-1064 (argument checks, the optional
// x2 step-scaling loop) while every pixel is produced on the GPU by srcnn_process_u8.
#include <cmath>
#include <new>

#include "../../include/libsrcnn_dropin.h"
#include "../../include/srcnn_amd.h"

namespace {
// the reference's two file-static settings (src/libsrcnn.cpp:91-92); unsynchronised there too
bool g_stepscale = false;
SRCNNFilterType g_filter = SRCNNF_Bicubic;

// one doSRCNN pass: allocate with new[] as the reference does, fill on the device
int one_pass(const unsigned char* src, unsigned w, unsigned h, unsigned d, float mul,
             unsigned char*& out, unsigned& outsz, unsigned char** conv, unsigned* convsz)
{
    const unsigned dw = (unsigned)((float)w * mul), dh = (unsigned)((float)h * mul);
    if (dw == 0 || dh == 0) return SRCNN_E_SCALE;
    const unsigned long long osz = (unsigned long long)dw * dh * d;
    if (osz > 0xffffffffULL) return SRCNN_E_OUTALLOC;       // outbuffsz is 32-bit in the API
    unsigned char* o = new (std::nothrow) unsigned char[osz];
    if (!o) return SRCNN_E_OUTALLOC;
    unsigned char* c = nullptr;
    const bool want_conv = conv && convsz;
    if (want_conv) {
        c = new (std::nothrow) unsigned char[(size_t)dw * dh];
        if (!c) { delete[] o; return SRCNN_E_CONVALLOC; }
    }
    const int rc = srcnn_process_u8(src, w, h, d, mul, (int)g_filter, o, c);
    if (rc != 0) { delete[] o; delete[] c; return rc; }
    out = o;
    outsz = (unsigned)osz;
    if (want_conv) { *conv = c; *convsz = dw * dh; }
    return 0;
}
}  // namespace

void ConfigureFilterSRCNN(SRCNNFilterType ftype, bool stepscale)
{
    g_filter = ftype;
    g_stepscale = stepscale;
}

int ProcessSRCNN(const unsigned char* refbuff, unsigned w, unsigned h, unsigned d, float multiply,
                 unsigned char*& outbuff, unsigned& outbuffsz, unsigned char** convbuff, unsigned* convbuffsz)
{
    if (refbuff == nullptr || w == 0 || h == 0 || d == 0) return -1;
    if ((float)w * multiply <= 0.f || (float)h * multiply <= 0.f) return -2;

    if (!g_stepscale) return one_pass(refbuff, w, h, d, multiply, outbuff, outbuffsz, convbuff, convbuffsz);

    // step scaling: repeated x2 passes, then whatever factor is left (src/libsrcnn.cpp:980-1061)
    int passes = (int)(multiply / 2.f);
    if (fmodf(multiply, 2.f) > 0.f) ++passes;
    const unsigned char* cur = refbuff;
    unsigned char* produced = nullptr;
    unsigned produced_sz = 0;
    unsigned cw = w, ch = h;
    int rc = -100;
    for (int p = 0; p < passes; ++p) {
        float f = 2.0f;
        const bool last = (p + 1 == passes);
        if (last) {
            f = ((float)w * multiply) / (float)cw;
            if (f == 0.f || f == 1.0f) break;
        }
        unsigned char* next = nullptr;
        unsigned next_sz = 0;
        rc = one_pass(cur, cw, ch, d, f, next, next_sz, last ? convbuff : nullptr, last ? convbuffsz : nullptr);
        if (cur != refbuff) delete[] cur;       // intermediate images are ours
        cur = nullptr;
        if (rc != 0) { produced = nullptr; break; }
        produced = next; produced_sz = next_sz;
        cur = next;
        if (passes > 1) { cw = (unsigned)((float)cw * f); ch = (unsigned)((float)ch * f); }
    }
    outbuff = produced;
    outbuffsz = produced_sz;
    return rc;
}

extern "C" void srcnn_delete_array(unsigned char* p) { delete[] p; }

extern "C" int srcnn_output_size(unsigned w, unsigned h, float multiply, int stepscale, unsigned* out_w, unsigned* out_h)
{
    if (w == 0 || h == 0) return -1;
    if ((float)w * multiply <= 0.f || (float)h * multiply <= 0.f) return -2;
    unsigned cw = w, ch = h;
    if (!stepscale) {
        cw = (unsigned)((float)w * multiply); ch = (unsigned)((float)h * multiply);
    } else {            // same pass structure as ProcessSRCNN above
        int passes = (int)(multiply / 2.f);
        if (fmodf(multiply, 2.f) > 0.f) ++passes;
        for (int p = 0; p < passes; ++p) {
            float f = 2.0f;
            if (p + 1 == passes) {
                f = ((float)w * multiply) / (float)cw;
                if (f == 0.f || f == 1.0f) break;
            }
            cw = (unsigned)((float)cw * f); ch = (unsigned)((float)ch * f);
        }
    }
    if (cw == 0 || ch == 0) return -2;
    if (out_w) *out_w = cw;
    if (out_h) *out_h = ch;
    return 0;
}
