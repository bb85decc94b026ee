// srcnntest -- command-line front end for the drop-in library, the counterpart of the reference's test
// harness (src/test.cpp:290-448 argument handling, :533-745 main) without its FLTK / libpng dependency:
// images are binary Netpbm (P6 RGB, P5 gray -> expanded to RGB as the reference's convImage does at
// src/test.cpp:56-80, P7 RGB_ALPHA).
//
//   srcnntest [--scale=<ratio>] [--step] [--filter=<0..4>] [--waitakey] [--devices=all|<id,id,...>] [--repeat=N]
//             [--sequence=N] source.ppm [output.ppm]
//
// --sequence=N (not in the reference, whose harness times one blocking call, src/test.cpp:653-672): after the ProcessSRCNN
// call, push the image N more times through the way a caller with a SEQUENCE of images should use the library -- page-locked
// source and result buffers (srcnn_host_alloc_pinned) and two asynchronous jobs in flight (srcnn_process_u8_begin / _wait) --
// check that every result equals the ProcessSRCNN bytes, and print the time per image.
//
// --waitakey pauses before exit like the reference's (src/test.cpp:735-742, there so that a human can watch the process'
// memory for leaks).  --devices selects the GPUs ONE ProcessSRCNN call may use (default: device 0; "all": every visible
// device -- the bands of a large image are dealt to them; a list may repeat an id).
//
// Writes <source>_resized.ppm (or the given output) and <source>_convolution.pgm (the truncated SRCNN Y
// plane), and prints the wall time of the ProcessSRCNN call like the reference ("Test Ok, took N ms.").
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/libsrcnn_dropin.h"
#include "../include/srcnn_amd.h"

namespace {

bool read_token(FILE* f, std::string& tok)
{
    tok.clear();
    int c;
    while ((c = fgetc(f)) != EOF) {
        if (c == '#') { while ((c = fgetc(f)) != EOF && c != '\n') {} continue; }
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { if (!tok.empty()) return true; continue; }
        tok.push_back((char)c);
    }
    return !tok.empty();
}

// returns depth (3 or 4) or 0 on failure; gray input is replicated to RGB
unsigned load_netpbm(const char* path, std::vector<unsigned char>& px, unsigned& w, unsigned& h)
{
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    std::string t;
    unsigned d = 0, src_d = 0, maxv = 255;
    if (!read_token(f, t)) { fclose(f); return 0; }
    if (t == "P6" || t == "P5") {
        src_d = (t == "P6") ? 3 : 1;
        std::string a, b, c;
        if (!read_token(f, a) || !read_token(f, b) || !read_token(f, c)) { fclose(f); return 0; }
        w = (unsigned)atoi(a.c_str()); h = (unsigned)atoi(b.c_str()); maxv = (unsigned)atoi(c.c_str());
        d = 3;
    } else if (t == "P7") {
        std::string key, val;
        while (read_token(f, key) && key != "ENDHDR") {
            if (!read_token(f, val)) break;
            if (key == "WIDTH") w = (unsigned)atoi(val.c_str());
            else if (key == "HEIGHT") h = (unsigned)atoi(val.c_str());
            else if (key == "DEPTH") src_d = (unsigned)atoi(val.c_str());
            else if (key == "MAXVAL") maxv = (unsigned)atoi(val.c_str());
        }
        d = (src_d == 4) ? 4 : 3;
    } else { fclose(f); return 0; }
    if (w == 0 || h == 0 || maxv != 255 || src_d == 0 || src_d == 2 || src_d > 4) { fclose(f); return 0; }
    std::vector<unsigned char> raw((size_t)w * h * src_d);
    const size_t got = fread(raw.data(), 1, raw.size(), f);
    fclose(f);
    if (got != raw.size()) return 0;
    if (src_d == d) { px.swap(raw); return d; }
    px.resize((size_t)w * h * d);
    for (size_t p = 0; p < (size_t)w * h; ++p)
        for (unsigned k = 0; k < 3; ++k) px[p * 3 + k] = raw[p * src_d + (src_d == 1 ? 0 : k)];
    return d;
}

bool save_netpbm(const std::string& path, const unsigned char* px, unsigned w, unsigned h, unsigned d)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    if (d == 1) fprintf(f, "P5\n%u %u\n255\n", w, h);
    else if (d == 3) fprintf(f, "P6\n%u %u\n255\n", w, h);
    else fprintf(f, "P7\nWIDTH %u\nHEIGHT %u\nDEPTH 4\nMAXVAL 255\nTUPLTYPE RGB_ALPHA\nENDHDR\n", w, h);
    const bool ok = fwrite(px, 1, (size_t)w * h * d, f) == (size_t)w * h * d;
    fclose(f);
    return ok;
}

std::string stem(const std::string& p)
{
    const size_t dot = p.find_last_of('.');
    return dot == std::string::npos ? p : p.substr(0, dot);
}

}  // namespace

int main(int argc, char** argv)
{
    float scale = 2.0f;                      // the reference's default image_multiply (src/test.cpp:288)
    bool step = false, waitakey = false;
    std::string devices;
    int repeat = 1;                          // --repeat=N: call ProcessSRCNN N times, report every wall time
    int sequence = 0;                        // --sequence=N: N more images through the asynchronous page-locked path
    SRCNNFilterType filt = SRCNNF_Bicubic;
    std::string src, dst;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a.rfind("--scale=", 0) == 0) { const float v = (float)atof(a.c_str() + 8); if (v > 0.f) scale = v; }
        else if (a.rfind("--step", 0) == 0) step = true;
        else if (a.rfind("--waitakey", 0) == 0) waitakey = true;
        else if (a.rfind("--devices=", 0) == 0) devices = a.substr(10);
        else if (a.rfind("--repeat=", 0) == 0) repeat = std::max(1, atoi(a.c_str() + 9));
        else if (a.rfind("--sequence=", 0) == 0) sequence = std::max(0, atoi(a.c_str() + 11));
        else if (a.rfind("--filter=", 0) == 0) {
            const int v = atoi(a.c_str() + 9);
            filt = (v >= 0 && v <= 4) ? (SRCNNFilterType)v : SRCNNF_Bicubic;
        } else if (src.empty()) src = a;
        else if (dst.empty()) dst = a;
    }
    if (src.empty()) {
        printf("usage: %s [--scale=<ratio>] [--step] [--filter=<0 nearest|1 bilinear|2 bicubic|3 lanczos3|4 b-spline>] "
               "[--waitakey] [--devices=all|<id,id,...>] [--repeat=N] [--sequence=N] source.(ppm|pgm|pam) [output]\n", argv[0]);
        return 0;
    }
    std::vector<unsigned char> img;
    unsigned w = 0, h = 0;
    const unsigned d = load_netpbm(src.c_str(), img, w, h);
    if (d == 0) { printf("- load failure: %s (binary P5/P6/P7, maxval 255 expected)\n", src.c_str()); return -1; }
    const bool alpha = (d == 4);
    if (dst.empty()) dst = stem(src) + "_resized" + (alpha ? ".pam" : ".ppm");
    const std::string cov = stem(src) + "_convolution.pgm";

    char dev[256] = "";
    int init_rc;
    if (devices.empty()) init_rc = srcnn_init(-1);           // env SRCNN_DEVICES (all | id,id,...) or device 0
    else if (devices == "all") init_rc = srcnn_init_devices(nullptr, 0);
    else {
        std::vector<int> ids;
        for (const char* p = devices.c_str(); *p;) {
            char* end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            ids.push_back((int)v);
            p = (*end == ',') ? end + 1 : end;
        }
        init_rc = ids.empty() ? -1 : srcnn_init_devices(ids.data(), (int)ids.size());
    }
    if (init_rc != 0) { printf("- device init failed: %s\n", srcnn_last_error()); return -200; }
    srcnn_device_name(dev, sizeof dev);
    printf("- device: %s, %d context(s)\n- Image loaded: %ux%ux%u, scaling ratio %.2f, filter %d%s\n", dev, srcnn_context_count(), w, h, d, scale, (int)filt,
           step ? ", step scaling" : "");

    ConfigureFilterSRCNN(filt, step);
    unsigned char* out = nullptr; unsigned outsz = 0;
    unsigned char* conv = nullptr; unsigned convsz = 0;
    int rc = 0;
    for (int it = 0; it < repeat; ++it) {
        delete[] out; delete[] conv; out = nullptr; conv = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        rc = ProcessSRCNN(img.data(), w, h, d, scale, out, outsz, &conv, &convsz);
        const auto t1 = std::chrono::steady_clock::now();
        if (rc != 0 || !out) { printf("- Failed, error code = %d (%s)\n", rc, srcnn_last_error()); return rc; }
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (repeat == 1) printf("- Test Ok, took %u ms.\n", (unsigned)ms);
        else printf("- Test Ok, took %.2f ms (call %d of %d).\n", ms, it + 1, repeat);
    }

    unsigned ow = 0, oh = 0;
    if (srcnn_output_size(w, h, scale, step ? 1 : 0, &ow, &oh) != 0 || (size_t)ow * oh * d != outsz) {
        printf("- Failed: unexpected output size %u\n", outsz);
        return -4;
    }
    int ret = 0;
    if (sequence > 0 && !step) {
        // the sequence path: one doSRCNN pass per image (no step scaling), buffers page-locked, two jobs in flight
        const size_t in_n = (size_t)w * h * d, out_n = (size_t)ow * oh * d;
        unsigned char* pin_in = static_cast<unsigned char*>(srcnn_host_alloc_pinned(in_n));
        unsigned char* pin_out[2] = {static_cast<unsigned char*>(srcnn_host_alloc_pinned(out_n)),
                                     static_cast<unsigned char*>(srcnn_host_alloc_pinned(out_n))};
        if (!pin_in || !pin_out[0] || !pin_out[1]) { printf("- sequence: page-locked allocation failed (%s)\n", srcnn_last_error()); ret = -5; }
        else {
            memcpy(pin_in, img.data(), in_n);
            void* job[2] = {nullptr, nullptr};
            bool same = true;
            int src_rc = 0;
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < sequence && !src_rc; ++i) {
                src_rc = srcnn_process_u8_begin(pin_in, w, h, d, scale, (int)filt, pin_out[i & 1], nullptr, &job[i & 1]);
                if (i && !src_rc) {
                    src_rc = srcnn_process_u8_wait(job[(i - 1) & 1]); job[(i - 1) & 1] = nullptr;
                    same = same && !src_rc && memcmp(pin_out[(i - 1) & 1], out, out_n) == 0;
                }
            }
            for (int k = 0; k < 2; ++k)
                if (job[k]) { const int r2 = srcnn_process_u8_wait(job[k]); if (!src_rc) src_rc = r2; same = same && !r2 && memcmp(pin_out[k], out, out_n) == 0; }
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (src_rc) { printf("- sequence failed, error code = %d (%s)\n", src_rc, srcnn_last_error()); ret = src_rc; }
            else printf("- sequence of %d images (page-locked buffers, two asynchronous jobs in flight): %.2f ms per image, results %s the ProcessSRCNN bytes\n",
                        sequence, ms / sequence, same ? "equal" : "DIFFER FROM");
            if (!src_rc && !same) ret = -6;
        }
        srcnn_host_free_pinned(pin_in); srcnn_host_free_pinned(pin_out[0]); srcnn_host_free_pinned(pin_out[1]);
    }
    if (!save_netpbm(dst, out, ow, oh, d)) { printf("- Failed to write %s\n", dst.c_str()); ret = -3; }
    else printf("- Saved %s (%ux%ux%u)\n", dst.c_str(), ow, oh, d);
    if (conv && convsz == ow * oh) {
        if (save_netpbm(cov, conv, ow, oh, 1)) printf("- Saved %s\n", cov.c_str());
    }
    delete[] out;
    delete[] conv;
    srcnn_shutdown();
    if (waitakey) {                         // src/test.cpp:735-742
        printf("- Input any key and enter to quit.\n");
        fflush(stdout);
        (void)getchar();
    }
    return ret;
}
