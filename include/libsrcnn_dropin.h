/*
 * libsrcnn_dropin.h -- the two C++-linkage entry points of rageworx/libsrcnn, as exported by
 * libsrcnn_amd.so.  Declarations only; signatures, enum values, default argument, ownership and
 * return codes follow the reference's public header (src/libsrcnn.h:35-54) so that a program
 * compiled against the reference header links against this library unchanged (the mangled names
 * are _Z20ConfigureFilterSRCNN15SRCNNFilterTypeb and _Z12ProcessSRCNNPKhjjjfRPhRjPS1_Pj).
 */
#ifndef LIBSRCNN_DROPIN_H
#define LIBSRCNN_DROPIN_H

#define LIBSRCNN_VERSION 0x00010A28 /* 0.1.10.40, the reference version this build mirrors */

typedef enum {
    SRCNNF_Nearest = 0,
    SRCNNF_Bilinear,
    SRCNNF_Bicubic,
    SRCNNF_Lanczos3,
    SRCNNF_Bspline
} SRCNNFilterType;

/* Process-global settings, exactly two (src/libsrcnn.cpp:91-92, 930-941). */
__attribute__((visibility("default")))
void ConfigureFilterSRCNN(SRCNNFilterType ftype, bool stepscale = false);

/* refbuff: interleaved 8-bit RGB (d=3) or RGBA (d=4), w*h*d bytes, borrowed.
 * outbuff: receives a new[]-allocated (w*m)*(h*m)*d image the caller must delete[];
 * convbuff/convbuffsz (both non-NULL to enable): new[]-allocated truncated SRCNN Y plane.
 * Returns 0, or -1 (NULL/zero arg), -2 (non-positive scaled size), -11/-12 (allocation),
 * -100 (nothing produced), or a SRCNN_E_* device error (<= -200) from srcnn_amd.h. */
__attribute__((visibility("default")))
int ProcessSRCNN(const unsigned char* refbuff, unsigned w, unsigned h, unsigned d, float multiply,
                 unsigned char*& outbuff, unsigned& outbuffsz, unsigned char** convbuff, unsigned* convbuffsz);

#endif
