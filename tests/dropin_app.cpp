// The stand-in for "a program that already uses rageworx/libsrcnn": written against the reference's public header
// (libsrcnn.h: default argument, reference parameters, enum), linked with -lsrcnn.  tests/test_dropin_binary.py builds it ONCE
// against the reference's own shared library and then runs the SAME binary on either library by switching LD_LIBRARY_PATH.
//   dropin_app                          -> prints the return codes of the argument checks and of one valid 4x4 call
//   dropin_app in.rgb w h d m out.rgb conv.y   -> one ProcessSRCNN call on a raw interleaved image, results written raw
// Test infrastructure.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "libsrcnn.h"

int main(int argc, char** argv)
{
    ConfigureFilterSRCNN(SRCNNF_Lanczos3);            // the header's default stepscale argument
    ConfigureFilterSRCNN(SRCNNF_Bicubic, false);
    if (argc < 8) {
        unsigned char px[4 * 4 * 3] = {0};
        unsigned char* out = nullptr;
        unsigned outsz = 0;
        const int a = ProcessSRCNN(nullptr, 4, 4, 3, 2.0f, out, outsz, nullptr, nullptr);
        const int b = ProcessSRCNN(px, 4, 4, 3, -1.0f, out, outsz, nullptr, nullptr);
        const int c = ProcessSRCNN(px, 4, 4, 3, 2.0f, out, outsz, nullptr, nullptr);
        std::printf("%d %d %d %u\n", a, b, c, outsz);
        delete[] out;
        return 0;
    }
    const unsigned w = (unsigned)atoi(argv[2]), h = (unsigned)atoi(argv[3]), d = (unsigned)atoi(argv[4]);
    const float m = (float)atof(argv[5]);
    std::vector<unsigned char> in((size_t)w * h * d);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(in.data(), 1, in.size(), f) != in.size()) return 2;
    fclose(f);
    unsigned char *out = nullptr, *conv = nullptr;
    unsigned outsz = 0, convsz = 0;
    const int rc = ProcessSRCNN(in.data(), w, h, d, m, out, outsz, &conv, &convsz);
    std::printf("%d %u %u\n", rc, outsz, convsz);
    if (rc != 0) return 0;
    f = fopen(argv[6], "wb"); fwrite(out, 1, outsz, f); fclose(f);
    f = fopen(argv[7], "wb"); fwrite(conv, 1, convsz, f); fclose(f);
    delete[] out;
    delete[] conv;
    return 0;
}
