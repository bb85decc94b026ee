// host_sanitize.cpp -- CPU-only sanitizer harness for the HOST code of the drop-in (the
// reference has latent out-of-bounds bugs of exactly this class, src/frawscale.cpp:185-193,249).
//
// Built by `make asan` (-fsanitize=address,undefined) and `make tsan` (-fsanitize=thread); no GPU, no HIP.
// What runs under the sanitizers:
//   * libsrcnn_amd/csrc/resample_table.hpp  (the product's contribution-table builder) for all five filters over
//     up-/down-scale ratios incl. 1-pixel axes, checked against the oracle's table bit for bit;
//   * libsrcnn_amd/csrc/dropin.cpp          (ProcessSRCNN / ConfigureFilterSRCNN: argument checks, the step-scaling
//     loop, new[] ownership of intermediates and results) with srcnn_process_u8 -- the one device call it makes --
//     replaced by a stand-in that forwards to the CPU oracle.  The stand-in exists only in this test binary;
//   * oracle/srcnn_oracle.c itself on odd plane shapes;
//   * concurrent ProcessSRCNN calls from 4 threads (the TSan target: the drop-in keeps no per-call state);
//   * libsrcnn_amd/csrc/srcnn_watchdog.hpp  (the deadline around every blocking RCCL call) hammered from 4 threads, some of
//     them toggling the timeout to 0 between calls: regions are exclusive, a no-op arm() never releases somebody else's region,
//     a region that outlives its deadline is marked (under the lock) and aborted exactly once, stale generations are ignored.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/libsrcnn_dropin.h"
#include "../../include/srcnn_amd.h"
#include "../../include/srcnn_amd_debug.h"
#include "../../libsrcnn_amd/csrc/resample_table.hpp"
#include "../../libsrcnn_amd/csrc/srcnn_watchdog.hpp"

extern "C" {
int oracle_axis_window(int filter, unsigned dst_len, unsigned src_len);
void oracle_axis_table(int filter, unsigned dst_len, unsigned src_len, int* left, int* right, double* w);
int oracle_y_path(const float* y, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter, float* out, float* up,
                  float* c1, float* c2);
int oracle_dosrcnn(const unsigned char* rgb, unsigned w, unsigned h, unsigned d, float mul, int filter, unsigned char* out,
                   unsigned char* conv_opt);
}

// ---- the stand-in for the device call (test binary only) ----
static std::atomic<int> g_calls{0};
extern "C" int srcnn_process_u8(const unsigned char* rgb, unsigned w, unsigned h, unsigned d, float multiply, int filter,
                                unsigned char* out, unsigned char* conv_opt)
{
    ++g_calls;
    if (d != 3 && d != 4) return SRCNN_E_UNSUPPORTED;
    return oracle_dosrcnn(rgb, w, h, d, multiply, filter, out, conv_opt);
}

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++g_fail; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

static void check_tables()
{
    const unsigned pairs[][2] = {{1, 1}, {2, 1}, {1, 2}, {7, 3}, {3, 7}, {50, 1}, {1, 50}, {34, 23}, {2160, 1080}, {11, 23},
                                 {512, 256}, {255, 256}, {257, 256}, {9, 600}};
    for (int f = 0; f < 5; ++f)
        for (auto& p : pairs) {
            const unsigned dst = p[0], src = p[1];
            const srcnn::AxisTable t = srcnn::build_axis_table(f, dst, src);
            const int win = oracle_axis_window(f, dst, src);
            CHECK(win == t.window, "window f=%d %u<-%u: %d vs %d", f, dst, src, t.window, win);
            std::vector<int> L(dst), R(dst);
            std::vector<double> Wt((size_t)dst * (win + 1), 0.0);
            oracle_axis_table(f, dst, src, L.data(), R.data(), Wt.data());
            for (unsigned u = 0; u < dst; ++u) {
                CHECK(t.first[u] >= 0 && t.first[u] + t.taps[u] <= (int)src && t.taps[u] >= 1 && t.taps[u] <= t.window,
                      "range f=%d %u<-%u u=%u: first %d taps %d", f, dst, src, u, t.first[u], t.taps[u]);
                CHECK(t.first[u] == L[u] && t.last[u] == R[u], "bounds f=%d %u<-%u u=%u", f, dst, src, u);
                CHECK(memcmp(&t.weight[(size_t)u * t.stride], &Wt[(size_t)u * (win + 1)], sizeof(double) * t.taps[u]) == 0,
                      "weights f=%d %u<-%u u=%u", f, dst, src, u);
            }
        }
}

static void check_oracle_shapes()
{
    const unsigned shapes[][2] = {{1, 1}, {17, 1}, {1, 13}, {31, 29}, {5, 64}};
    for (auto& s : shapes) {
        const unsigned w = s[0], h = s[1];
        std::vector<float> y((size_t)w * h), out((size_t)4 * w * h);
        for (size_t i = 0; i < y.size(); ++i) y[i] = (float)((i * 37) % 256);
        CHECK(oracle_y_path(y.data(), w, h, 2 * w, 2 * h, 2, out.data(), nullptr, nullptr, nullptr) == 0, "y_path %ux%u", w, h);
        for (float v : out) CHECK(std::isfinite(v) && v >= 0.f && v <= 255.f, "range");
    }
}

static std::vector<unsigned char> image(unsigned w, unsigned h, unsigned d, unsigned seed)
{
    std::vector<unsigned char> v((size_t)w * h * d);
    unsigned x = seed * 2654435761u + 1;
    for (auto& b : v) { x = x * 1664525u + 1013904223u; b = (unsigned char)(x >> 24); }
    return v;
}

// what ProcessSRCNN must return for (img, mul, step): the reference's pass structure replayed through the oracle
static std::vector<unsigned char> expected(const std::vector<unsigned char>& img, unsigned w, unsigned h, unsigned d, float mul,
                                           bool step, int filter, unsigned& ow, unsigned& oh)
{
    std::vector<unsigned char> cur = img;
    unsigned cw = w, ch = h;
    auto pass = [&](float f) {
        const unsigned nw = (unsigned)((float)cw * f), nh = (unsigned)((float)ch * f);
        std::vector<unsigned char> nxt((size_t)nw * nh * d);
        oracle_dosrcnn(cur.data(), cw, ch, d, f, filter, nxt.data(), nullptr);
        cur.swap(nxt); cw = nw; ch = nh;
    };
    if (!step) pass(mul);
    else {
        int passes = (int)(mul / 2.f);
        if (fmodf(mul, 2.f) > 0.f) ++passes;
        for (int p = 0; p < passes; ++p) {
            float f = 2.f;
            if (p + 1 == passes) { f = ((float)w * mul) / (float)cw; if (f == 0.f || f == 1.f) break; }
            pass(f);
        }
    }
    ow = cw; oh = ch;
    return cur;
}

static void check_dropin()
{
    unsigned char* out = nullptr; unsigned osz = 0;
    const std::vector<unsigned char> tiny = image(4, 4, 3, 1);
    CHECK(ProcessSRCNN(nullptr, 4, 4, 3, 2.f, out, osz, nullptr, nullptr) == -1, "NULL");
    CHECK(ProcessSRCNN(tiny.data(), 0, 4, 3, 2.f, out, osz, nullptr, nullptr) == -1, "w=0");
    CHECK(ProcessSRCNN(tiny.data(), 4, 4, 0, 2.f, out, osz, nullptr, nullptr) == -1, "d=0");
    CHECK(ProcessSRCNN(tiny.data(), 4, 4, 3, 0.f, out, osz, nullptr, nullptr) == -2, "mul=0");
    CHECK(ProcessSRCNN(tiny.data(), 4, 4, 3, -1.f, out, osz, nullptr, nullptr) == -2, "mul<0");
    CHECK(ProcessSRCNN(tiny.data(), 4, 4, 3, 0.1f, out, osz, nullptr, nullptr) == -2, "scaled size 0");
    CHECK(out == nullptr && g_calls == 0, "no device call / no allocation on the error paths");

    struct Case { unsigned w, h, d; float mul; bool step; int filter; };
    const Case cases[] = {{24, 20, 3, 2.f, false, 2}, {24, 20, 4, 2.f, false, 2}, {13, 9, 3, 1.5f, false, 3}, {10, 7, 4, 3.f, false, 1},
                          {12, 10, 3, 4.f, true, 2}, {12, 10, 3, 3.f, true, 2}, {9, 8, 4, 2.5f, true, 4}, {9, 8, 3, 2.f, true, 0},
                          {1, 1, 3, 2.f, false, 2}, {1, 5, 4, 6.f, true, 2}};
    for (const Case& c : cases) {
        const std::vector<unsigned char> img = image(c.w, c.h, c.d, c.w * 31 + c.h);
        ConfigureFilterSRCNN((SRCNNFilterType)c.filter, c.step);
        unsigned char *o = nullptr, *cv = nullptr; unsigned os = 0, cs = 0;
        const int rc = ProcessSRCNN(img.data(), c.w, c.h, c.d, c.mul, o, os, &cv, &cs);
        unsigned ow = 0, oh = 0, qw = 0, qh = 0;
        const std::vector<unsigned char> want = expected(img, c.w, c.h, c.d, c.mul, c.step, c.filter, ow, oh);
        CHECK(rc == 0, "rc %d for %ux%ux%u x%.1f step=%d", rc, c.w, c.h, c.d, c.mul, (int)c.step);
        CHECK(srcnn_output_size(c.w, c.h, c.mul, c.step, &qw, &qh) == 0 && qw == ow && qh == oh, "output_size %ux%u vs %ux%u", qw, qh, ow, oh);
        CHECK(os == ow * oh * c.d && cs == ow * oh, "sizes %u %u", os, cs);
        CHECK(o && memcmp(o, want.data(), want.size()) == 0, "bytes for %ux%ux%u x%.1f step=%d", c.w, c.h, c.d, c.mul, (int)c.step);
        srcnn_delete_array(o);          // delete[]: ASan's alloc-dealloc-mismatch check pins the new[] contract
        delete[] cv;
        // without the conv-Y out-params
        o = nullptr; os = 0;
        CHECK(ProcessSRCNN(img.data(), c.w, c.h, c.d, c.mul, o, os, nullptr, nullptr) == 0 && os == ow * oh * c.d, "no-conv call");
        delete[] o;
    }
    ConfigureFilterSRCNN(SRCNNF_Bicubic, false);
    // d outside {3,4}: the stand-in (like the product) refuses; nothing may leak
    const std::vector<unsigned char> gray = image(6, 6, 1, 3);
    out = nullptr; osz = 0;
    CHECK(ProcessSRCNN(gray.data(), 6, 6, 1, 2.f, out, osz, nullptr, nullptr) == SRCNN_E_UNSUPPORTED && out == nullptr, "d=1");
}

static void check_threads()
{
    ConfigureFilterSRCNN(SRCNNF_Bicubic, false);
    std::vector<std::thread> th;
    std::atomic<int> bad{0};
    for (int t = 0; t < 4; ++t)
        th.emplace_back([t, &bad] {
            const unsigned w = 10 + 3 * t, h = 8 + t, d = 3 + (t & 1);
            const std::vector<unsigned char> img = image(w, h, d, 100 + t);
            unsigned ow, oh;
            const std::vector<unsigned char> want = expected(img, w, h, d, 2.f, false, 2, ow, oh);
            for (int it = 0; it < 8; ++it) {
                unsigned char *o = nullptr, *cv = nullptr; unsigned os = 0, cs = 0;
                if (ProcessSRCNN(img.data(), w, h, d, 2.f, o, os, &cv, &cs) != 0 || os != want.size() || memcmp(o, want.data(), os) != 0) ++bad;
                delete[] o; delete[] cv;
                unsigned char* e = nullptr; unsigned es = 0;
                if (ProcessSRCNN(nullptr, w, h, d, 2.f, e, es, nullptr, nullptr) != -1) ++bad;      // argument-check path
            }
        });
    for (auto& t : th) t.join();
    CHECK(bad == 0, "%d concurrent calls went wrong", bad.load());
}

static void check_watchdog()
{
    // (everything the watchdog's detached thread can touch lives for ever, as in the product, where the watchdog is a leaked singleton)
    static std::atomic<int> marks{0}, aborts{0}, inside{0}, overlap{0};
    static std::atomic<unsigned> current_gen{1};
    static srcnn::Watchdog& wd = *new srcnn::Watchdog([](unsigned gen) { if (gen == current_gen.load()) ++marks; },
                                                      [](void*, unsigned gen) { if (gen == current_gen.load()) ++aborts; });
    std::atomic<int> timeout{30};
    std::atomic<int> fired_seen{0}, slow_regions{0};
    auto worker = [&](int id) {
        for (int it = 0; it < 200; ++it) {
            if (id == 3 && (it % 7) == 0) timeout = timeout.load() ? 0 : 30;          // a thread that switches the deadline off and on
            int dummy = 0;
            const bool armed = wd.arm(&dummy, current_gen.load(), timeout.load());
            if (armed) {
                if (inside.fetch_add(1) != 0) ++overlap;                              // armed regions must be exclusive
                const bool slow = id == 0 && (it % 50) == 49;                         // a "blocked RCCL call": outlives its deadline
                if (slow) { ++slow_regions; std::this_thread::sleep_for(std::chrono::milliseconds(80)); }
                inside.fetch_sub(1);
            }
            if (wd.disarm(armed)) ++fired_seen;
        }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < 4; ++i) th.emplace_back(worker, i);
    for (auto& t : th) t.join();
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    CHECK(overlap.load() == 0, "watchdog: %d overlapping armed regions", overlap.load());
    CHECK(fired_seen.load() == slow_regions.load(), "watchdog: %d regions outlived their deadline, %d saw it", slow_regions.load(), fired_seen.load());
    CHECK(marks.load() == slow_regions.load() && aborts.load() == slow_regions.load(), "watchdog: marks %d aborts %d for %d slow regions",
          marks.load(), aborts.load(), slow_regions.load());
    // a stale generation: the region was armed for communicator generation 1, the communicator is replaced while it blocks
    int dummy = 0;
    const int m0 = marks.load(), a0 = aborts.load();
    const bool armed = wd.arm(&dummy, 1, 20);
    current_gen = 2;
    std::this_thread::sleep_for(std::chrono::milliseconds(60));
    CHECK(wd.disarm(armed), "watchdog: the stale region's deadline still fires for its owner");
    CHECK(marks.load() == m0 && aborts.load() == a0, "watchdog: a stale generation must not touch the new communicator");
}

int main()
{
    check_tables();
    check_oracle_shapes();
    check_dropin();
    check_threads();
    check_watchdog();
    if (g_fail) { fprintf(stderr, "host_sanitize: %d check(s) failed\n", g_fail); return 1; }
    printf("host_sanitize: all checks passed (%d stand-in device calls)\n", g_calls.load());
    return 0;
}
