// host_out_probe -- how should 100 MB of results reach a caller's freshly new[]-ed (never touched) buffer?
// ProcessSRCNN must hand back new[] memory (the caller delete[]s it, src/libsrcnn.cpp:874-887), so the pages of the
// destination fault in on first touch.  Measures, for one 7680x4320x3 result:
//   a  memcpy from page-locked staging into the fresh buffer, 1 / 4 / 8 threads            (round-2 path)
//   b  the same after madvise(MADV_HUGEPAGE)
//   c  MADV_POPULATE_WRITE prefault (1 / 4 / 8 threads), then memcpy
//   d  hipHostRegister of the fresh buffer (1 / 4 chunks in threads) + D2H straight into it + hipHostUnregister
// Build: make ubench.  Run on the GPU box.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class F>
static void par(unsigned nt, size_t n, F f)
{
    std::vector<std::thread> th;
    const size_t per = ((n / nt) + 4095) & ~size_t(4095);
    for (unsigned t = 0; t < nt; ++t) {
        const size_t off = (size_t)t * per;
        if (off >= n) break;
        const size_t len = std::min(per, n - off);
        th.emplace_back([=] { f(off, len); });
    }
    for (auto& t : th) t.join();
}

int main()
{
    const size_t n = (size_t)7680 * 4320 * 3;
    unsigned char* pin = nullptr;
    unsigned char* dev = nullptr;
    if (hipHostMalloc((void**)&pin, n, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&dev, n) != hipSuccess) return 1;
    memset(pin, 7, n);
    hipMemset(dev, 9, n);
    hipDeviceSynchronize();
    auto fresh = [&] { return new unsigned char[n]; };
    for (unsigned nt : {1u, 4u, 8u, 16u}) {
        unsigned char* d = fresh();
        double t0 = now_ms();
        par(nt, n, [&](size_t off, size_t len) { memcpy(d + off, pin + off, len); });
        printf("a  memcpy into fresh new[]            %2u threads: %6.2f ms\n", nt, now_ms() - t0);
        t0 = now_ms();
        par(nt, n, [&](size_t off, size_t len) { memcpy(d + off, pin + off, len); });
        printf("   memcpy again (pages present)       %2u threads: %6.2f ms\n", nt, now_ms() - t0);
        delete[] d;
    }
    for (unsigned nt : {1u, 4u, 8u}) {
        unsigned char* d = fresh();
        unsigned char* al = (unsigned char*)(((uintptr_t)d + 4095) & ~(uintptr_t)4095);
        const int rc = madvise(al, n - (al - d) - 4096, MADV_HUGEPAGE);
        double t0 = now_ms();
        par(nt, n, [&](size_t off, size_t len) { memcpy(d + off, pin + off, len); });
        printf("b  MADV_HUGEPAGE (rc %d) then memcpy   %2u threads: %6.2f ms\n", rc, nt, now_ms() - t0);
        delete[] d;
    }
    for (int huge = 0; huge < 2; ++huge)
        for (unsigned nt : {1u, 4u, 8u}) {
            unsigned char* d = fresh();
            unsigned char* al = (unsigned char*)(((uintptr_t)d + 4095) & ~(uintptr_t)4095);
            const size_t aln = (n - (al - d)) & ~size_t(4095);
            if (huge) madvise(al, aln, MADV_HUGEPAGE);
            int bad = 0;
            double t0 = now_ms();
            par(nt, aln, [&](size_t off, size_t len) { if (madvise(al + off, len, MADV_POPULATE_WRITE) != 0) bad = 1; });
            const double tp = now_ms() - t0;
            t0 = now_ms();
            par(8, n, [&](size_t off, size_t len) { memcpy(d + off, pin + off, len); });
            printf("c  POPULATE_WRITE%s %2u threads: %6.2f ms%s, then memcpy (8 thr) %6.2f ms\n", huge ? " +HUGEPAGE" : "          ", nt, tp,
                   bad ? " (FAILED)" : "", now_ms() - t0);
            delete[] d;
        }
    for (unsigned chunks : {1u, 4u, 8u}) {
        unsigned char* d = fresh();
        double t0 = now_ms();
        int bad = 0;
        par(chunks, n, [&](size_t off, size_t len) { if (hipHostRegister(d + off, len, hipHostRegisterDefault) != hipSuccess) bad = 1; });
        const double tr = now_ms() - t0;
        t0 = now_ms();
        hipMemcpy(d, dev, n, hipMemcpyDeviceToHost);
        const double tc = now_ms() - t0;
        t0 = now_ms();
        par(chunks, n, [&](size_t off, size_t len) { hipHostUnregister(d + off); (void)len; });
        printf("d  hipHostRegister in %u chunk(s): %6.2f ms%s, D2H straight into it %6.2f ms, unregister %6.2f ms  (byte %d)\n", chunks, tr,
               bad ? " (FAILED)" : "", tc, now_ms() - t0, d[n / 2]);
        delete[] d;
    }
    {   // reference points: D2H into page-locked staging, and into pageable memory directly
        double t0 = now_ms();
        hipMemcpy(pin, dev, n, hipMemcpyDeviceToHost);
        printf("   D2H into page-locked staging: %6.2f ms\n", now_ms() - t0);
        unsigned char* d = fresh();
        t0 = now_ms();
        hipMemcpy(d, dev, n, hipMemcpyDeviceToHost);
        printf("   D2H into fresh pageable new[]: %6.2f ms\n", now_ms() - t0);
        delete[] d;
    }
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    if (f) { char b[128] = ""; if (fgets(b, sizeof b, f)) printf("THP: %s", b); fclose(f); }
    return 0;
}
