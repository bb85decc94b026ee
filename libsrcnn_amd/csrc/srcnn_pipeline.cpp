// srcnn_pipeline.cpp -- the host-pointer side of the C ABI: frames and images that live in HOST memory go through
// copy / compute / copy pipelines here, on one context or dealt over all contexts of the process (node level).
//
//   srcnn_y_upscale2x_f32_stream   stream of planar-float Y frames (BASELINE config #5): two device slots per context,
//                                  one hipGraph per slot; with several contexts the frames are dealt to them in
//                                  contiguous chunks, one worker thread per context.
//   srcnn_process_u8               the ProcessSRCNN surface (src/libsrcnn.cpp:628-923 = one doSRCNN pass): the output is
//                                  produced in horizontal bands; per band: stage the source rows the band needs (host
//                                  memcpy -> page-locked staging -> H2D), kernels, D2H into page-locked staging, fan-out to
//                                  the caller's buffer.  With several contexts the bands are dealt to the node's devices.
//   srcnn_y_upscale2x_f32_node_dev ONE device-resident frame tiled over all contexts of this process: each context pulls
//                                  the source rows its band needs from the root device (hipMemcpyPeerAsync), computes the
//                                  band in sub-bands, and pushes every finished sub-band to the root while the next one
//                                  computes (the single-process counterpart of the RCCL band gather).
//
// Every copy/kernel dependency that involves a copy engine is resolved on the HOST by a helper thread: on this runtime a copy
// that waits device-side on another queue's event does not overlap that queue's kernels (profiles/r02_stream_overlap.txt).
// The helpers sleep on condition variables and POLL device events (wait_event / wait_stream, srcnn_host.hpp): the runtime's
// own waits hold a core for the whole wait, hipEventBlockingSync or not (profiles/r03_wait_cost.txt).
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "srcnn_host.hpp"

namespace srcnn {
namespace {

// srcnn_debug_process_phases: where the time of the calling thread's last large srcnn_process_u8 / ProcessSRCNN went (share 0
// of the call; milliseconds since the share's entry).  Written by the thread that runs the share -- the caller's own thread for
// an image that is not dealt over several contexts.
struct PhaseRecord { double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int n = 0; };
thread_local PhaseRecord g_phases;

// srcnn_debug_stream_mode: how the frames of the process's last srcnn_y_upscale2x_f32_stream call were launched
std::atomic<unsigned> g_stream_graph_frames{0}, g_stream_plain_frames{0};
std::atomic<int> g_stream_fell_back{0};

double process_cpu_seconds()
{
    timespec ts;
    return clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts) == 0 ? (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec : 0.0;
}

bool is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    const bool yes = hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    return yes;
}

// The caller's result buffers are fresh new[] blocks (src/libsrcnn.cpp:874-887 hands back new[] memory) whose pages fault
// in on first touch; with 4 KB pages the fan-out memcpy is fault-bound (100 MB: 6.6-8 ms on 4-8 threads), with transparent
// huge pages it runs at memory speed (1.5 ms) -- profiles/r03_host_out_probe.txt.  A hint only: harmless where THP is off.
void hint_huge_pages(void* p, size_t n)
{
    if (!settings().thp) return;                               // A/B runs: a host with THP = never
    const uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 4095) & ~uintptr_t(4095);
    const uintptr_t e = (reinterpret_cast<uintptr_t>(p) + n) & ~uintptr_t(4095);
    if (e > a + (4u << 20)) (void)madvise(reinterpret_cast<void*>(a), e - a, MADV_HUGEPAGE);
}

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
// Fault the pages of a result buffer in BEFORE the fan-out needs them: the first few milliseconds of a large call belong to
// the GPU alone, so a helper thread spends them on the page faults (and the zeroing the kernel does per fresh page) that the
// fan-out memcpy would otherwise pay band by band, the last ones in the call's tail.  MADV_POPULATE_WRITE (Linux 5.14) does
// not touch the data, so it is safe beside a memcpy that already writes the same range; where it is not available nothing
// is done (the memcpy faults the pages itself, as before).
void prefault_range(uintptr_t a, uintptr_t e)
{
    const size_t chunk = 8u << 20;
    for (uintptr_t q = a; q < e; q += chunk)
        if (madvise(reinterpret_cast<void*>(q), std::min<size_t>(chunk, e - q), MADV_POPULATE_WRITE) != 0) return;
}

// `threads` workers, each on a contiguous share of the range, in address order within a share; the first share is the one
// the fan-out needs first.  One worker populates 100 MB in 3.8 ms into huge pages and 6 ms into 4 KB pages, four in 1.3 / 2.5 ms
// (profiles/r03_host_out_probe.txt): with 4 KB pages a lone worker is only barely ahead of the device.
void prefault_pages(void* p, size_t n, int threads = 1)
{
    const uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 4095) & ~uintptr_t(4095);
    const uintptr_t e = (reinterpret_cast<uintptr_t>(p) + n) & ~uintptr_t(4095);
    if (e <= a) return;
    threads = std::max(1, std::min<int>(threads, (int)((e - a) >> 24) + 1));      // at least 16 MB per worker
    if (threads == 1) { prefault_range(a, e); return; }
    const uintptr_t share = (((e - a) / threads) + 4095) & ~uintptr_t(4095);
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) {
        const uintptr_t b0 = std::min(e, a + t * share), b1 = std::min(e, b0 + share);
        try { pool.emplace_back([=] { prefault_range(b0, t + 1 == threads ? e : b1); }); }
        catch (...) { prefault_range(b0, t + 1 == threads ? e : b1); }
    }
    prefault_range(a, std::min(e, a + share));
    for (auto& t : pool) t.join();
}

constexpr unsigned kBlockingEvent = hipEventDisableTiming | hipEventBlockingSync;   // (the flag is honoured only under hipDeviceScheduleBlockingSync; our waits poll)

// Start a helper thread; false (and nothing started) if the system refuses -- callers then run the work inline.
template <class F>
bool try_thread(std::thread& t, F&& f)
{
    try { t = std::thread(std::forward<F>(f)); return true; }
    catch (...) { return false; }
}

// ------------------------------------------------------------------------------------------------------------------
// Frame stream on ONE context.  in/out are already page-locked (or pageable: the copies then just do not overlap).
// ------------------------------------------------------------------------------------------------------------------
int stream_on_ctx(Ctx& cx, const float* in, unsigned w, unsigned h, unsigned nframes, float* out, int use_graph, int mode,
                  bool in_locked, bool out_locked)
{
    int rc = bind(cx);
    if (rc) return rc;
    const size_t in_n = (size_t)w * h, out_n = in_n * 4;
    const size_t in_b = in_n * sizeof(float), out_b = out_n * sizeof(float);
    std::lock_guard<std::mutex> slk(cx.stream_mu);
    const int nslots = nframes > 1 ? 2 : 1;
    for (int i = 0; i < nslots && !rc; ++i) {
        StreamSlot& sl = cx.slots[i];
        if (!sl.st && hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking) != hipSuccess) rc = fail(SRCNN_E_HIP, "stream create");
        if (!rc && !sl.cst && hipStreamCreateWithFlags(&sl.cst, hipStreamNonBlocking) != hipSuccess) rc = fail(SRCNN_E_HIP, "stream create");
        for (hipEvent_t* e : {&sl.e_in, &sl.e_k, &sl.e_out})
            if (!rc && !*e && hipEventCreateWithFlags(e, kBlockingEvent) != hipSuccess) rc = fail(SRCNN_E_HIP, "event create");
        if (!rc && (sl.gw != w || sl.gh != h || sl.gmode != mode)) sl.graph_verdict = 0;
        if (!rc && (sl.gw != w || sl.gh != h || sl.gmode != mode)) {     // shape or mode changed: drop the graph first,
            if (sl.exec) { (void)wait_stream(cx.slots[0].st); (void)hipGraphExecDestroy(sl.exec); sl.exec = nullptr; }
            sl.ws.frozen = false;                                          // then its buffers may move again
            sl.graph_tables.clear();
            sl.tables.clear();
            sl.gw = w; sl.gh = h; sl.gmode = mode; sl.uses = 0;
        }
        if (!rc) rc = grow(sl.din, sl.din_n, in_n);
        if (!rc) rc = grow(sl.dout, sl.dout_n, out_n);
    }
    if (rc) return rc;
    // Pipeline.  Copies run on the slots' copy-only streams and every copy/kernel dependency that involves a copy is
    // resolved on the HOST (see the file header; measured with tools/hs_probe.py, 4K frames: 12.2-12.4 ms per frame with
    // the D2H on the kernel stream or behind hipStreamWaitEvent, 10.8 ms when the host waits for the kernels and then
    // queues the copy on an idle stream).  A copier thread waits for frame f's kernels and then issues its D2H; the main
    // thread waits for the slot's previous D2H before it reuses the slot.  Both sleep or poll; neither spins.
    Handoff launched;      // frames whose kernels have been queued (e_k recorded)
    Handoff copied;        // frames whose D2H has been queued (e_out recorded)
    std::atomic<int> copy_err{0};
    std::mutex capture_mu; // the copier's event queries stay out of the kernel stream's graph captures
    auto copy_frame = [&](unsigned f) {
        StreamSlot& sl = cx.slots[f % nslots];
        // (a buffer that could not be page-locked is never handed to the runtime as it is: HostBounce)
        if (wait_event(sl.e_k, &capture_mu) != hipSuccess ||
            (out_locked ? hipMemcpyAsync(out + f * out_n, sl.dout, out_b, hipMemcpyDeviceToHost, sl.cst) != hipSuccess
                        : copy_d2h_any(cx, out + f * out_n, sl.dout, out_b, sl.cst) != SRCNN_OK) ||
            hipEventRecord(sl.e_out, sl.cst) != hipSuccess) copy_err = 1;
        copied.publish(f + 1);
    };
    std::thread copier;
    const bool threaded = try_thread(copier, [&] {
        (void)hipSetDevice(cx.device);
        for (unsigned f = 0; f < nframes; ++f) {
            if (!launched.wait_for(f)) return;
            copy_frame(f);
        }
    });
    hipStream_t ks = cx.slots[0].st;       // ALL kernels go to one stream: frames back to back, never two frames' kernels
                                           // sharing the chip (that costs more than it overlaps: the persistent layer-1+2
                                           // kernel partitions its tiles over the workgroups it expects to be resident)
    // use_graph: 0 = plain launches; 2 = one hipGraph per slot, replayed per frame, whatever it costs; 1 = the same, KEPT ONLY
    // IF IT IS CHEAP: the process's CPU time over the first replayed frames is held against their wall time, and when a replay
    // costs more than SRCNN_GRAPH_MAX_CPU_PCT of a frame (on ROCm 7.2 a runtime thread spins from graph launch to completion:
    // 9.9 ms of CPU per 9.5 ms 4K frame, against 0.6 ms for the plain launches, at the same throughput -- profiles/r05_bench.json)
    // the graphs are retired and the stream goes on with plain launches.  The verdict is remembered per slot and shape.
    constexpr unsigned kProbeFrames = 4;
    const long max_pct = settings().graph_max_cpu_pct;
    unsigned probe_replays = 0;
    double probe_cpu0 = 0.0;
    std::chrono::steady_clock::time_point probe_t0;
    for (unsigned f = 0; f < nframes && !rc; ++f) {
        StreamSlot& sl = cx.slots[f % nslots];
        if (use_graph == 1 && max_pct > 0 && probe_replays == kProbeFrames && cx.slots[0].graph_verdict == 0) {
            // the slot of this frame has been waited for below in the previous round: kProbeFrames replays lie (almost) behind us
            const double cpu = process_cpu_seconds() - probe_cpu0;
            const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - probe_t0).count();
            const int verdict = (wall > 0 && cpu * 100.0 > wall * (double)max_pct) ? 2 : 1;
            for (int i = 0; i < nslots; ++i) cx.slots[i].graph_verdict = verdict;
            if (verdict == 2) g_stream_fell_back = 1;
        }
        const bool graph_now = use_graph == 2 || (use_graph == 1 && sl.graph_verdict != 2) || (use_graph != 0 && use_graph != 1 && use_graph != 2);
        Call c;
        c.cx = &cx; c.s = ks; c.ws = &sl.ws; c.mode = mode; c.hold = &sl.tables;
        if (f >= (unsigned)nslots) {
            // the slot's previous frame: its kernels are done (the copier saw e_k) once its D2H has been queued; wait for
            // that D2H to finish before din / dout are reused
            if (threaded && !copied.wait_for(f - nslots)) { rc = fail(SRCNN_E_HIP, "frame stream cancelled"); break; }
            if (wait_event(sl.e_out) != hipSuccess) { rc = fail(SRCNN_E_HIP, "D2H"); break; }
        }
        // frame in: also resolved on the host (the previous frame's kernels keep the device busy meanwhile)
        if ((in_locked ? hipMemcpyAsync(sl.din, in + f * in_n, in_b, hipMemcpyHostToDevice, sl.cst) != hipSuccess
                       : copy_h2d_any(cx, sl.din, in + f * in_n, in_b, sl.cst) != SRCNN_OK) ||
            hipEventRecord(sl.e_in, sl.cst) != hipSuccess || wait_event(sl.e_in) != hipSuccess) {
            rc = fail(SRCNN_E_HIP, "H2D"); break;
        }
        if (graph_now && sl.uses >= 1 && !sl.exec) {
            // The slot has run this shape eagerly once: tables and workspaces exist, so the kernel sequence
            // can be captured without any allocation inside the capture.  The graph's table references are kept apart
            // from the eager runs' (which trim theirs), for exactly as long as the graph lives.
            hipGraph_t graph = nullptr;
            sl.ws.frozen = true;
            c.timing = false;              // event pairs cannot be timed inside a capture
            c.hold = &sl.graph_tables;
            {
                std::lock_guard<std::mutex> cap(capture_mu);
                if (hipStreamBeginCapture(ks, hipStreamCaptureModeThreadLocal) != hipSuccess) rc = fail(SRCNN_E_HIP, "begin capture");
                if (!rc) rc = y_path_frame(c, sl.din, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, sl.dout);
                if (hipStreamEndCapture(ks, &graph) != hipSuccess && !rc) rc = fail(SRCNN_E_HIP, "end capture");
            }
            c.timing = true;
            c.hold = &sl.tables;
            if (!rc && hipGraphInstantiate(&sl.exec, graph, nullptr, nullptr, 0) != hipSuccess) rc = fail(SRCNN_E_HIP, "graph instantiate");
            if (graph) (void)hipGraphDestroy(graph);
            if (rc) { sl.ws.frozen = false; sl.graph_tables.clear(); break; }
        }
        if (graph_now && sl.exec) {
            if (probe_replays == 0) { probe_cpu0 = process_cpu_seconds(); probe_t0 = std::chrono::steady_clock::now(); }
            if (hipGraphLaunch(sl.exec, ks) != hipSuccess) { rc = fail(SRCNN_E_HIP, "graph launch"); break; }
            ++probe_replays;
            ++g_stream_graph_frames;
        } else {
            ++g_stream_plain_frames;
            if (sl.exec) {
                // an eager call (use_graph == 0) on a slot that still holds a captured graph of this shape: retire the graph
                // first.  Its workspace is frozen (pointers baked in), so an eager run that needs more scratch -- a larger
                // srcnn_set_workspace_limit since the capture -- could not grow it; and nothing should keep a graph alive
                // that the caller no longer asks for.  The next use_graph call captures again after one eager frame.
                (void)wait_stream(ks);
                (void)hipGraphExecDestroy(sl.exec);
                sl.exec = nullptr;
                sl.ws.frozen = false;
                sl.graph_tables.clear();
            }
            if (sl.tables.size() > 16) sl.tables.clear();   // eager runs re-take their references every frame
            rc = y_path_frame(c, sl.din, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, sl.dout);
            if (rc) break;
        }
        if (hipEventRecord(sl.e_k, ks) != hipSuccess) { rc = fail(SRCNN_E_HIP, "event record"); break; }
        ++sl.uses;
        if (threaded) launched.publish(f + 1);
        else copy_frame(f);
    }
    if (rc) launched.cancel();             // the copier stops at the first frame that was never launched
    if (threaded) copier.join();
    for (int i = 0; i < nslots; ++i) {
        if (cx.slots[i].st) (void)wait_stream(cx.slots[i].st);
        if (cx.slots[i].cst) (void)wait_stream(cx.slots[i].cst);
    }
    if (!rc && copy_err) rc = fail(SRCNN_E_HIP, "a device-to-host copy of the frame stream failed");
    return rc;
}

// ------------------------------------------------------------------------------------------------------------------
// ProcessSRCNN surface.
// ------------------------------------------------------------------------------------------------------------------
// One link of the chain that orders the KERNELS of consecutive asynchronous jobs on a device (srcnn_process_u8_begin): a job
// records `ev` on its compute stream once its last band's kernels are queued, its successor makes its own compute stream wait
// for `ev` before its first kernel.  The successor's stage-in copies do not wait, and the predecessor's copy-out and fan-out
// run beside the successor's kernels: the device goes from the last kernel of one image to the first of the next without
// idling and without two images' kernels sharing it.  `done` (under m) says the link is settled: recorded, or given up.
struct AsyncLink {
    std::mutex m;
    std::condition_variable cv;
    bool done = false, valid = false;
    bool leased = false;        // the job holds its lane: lanes are taken in chain order, or later jobs could take them all and
                                // wait for a predecessor that waits for a lane
    hipEvent_t ev = nullptr;
    int device = 0;
    void settle(bool ok)
    {
        { std::lock_guard<std::mutex> lk(m); if (done) return; done = true; leased = true; valid = ok; }
        cv.notify_all();
    }
    void mark_leased() { { std::lock_guard<std::mutex> lk(m); leased = true; } cv.notify_all(); }
    void wait_leased() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return leased; }); }
    ~AsyncLink()
    {
        if (!ev) return;
        int cur = -1;                              // (the last reference may go on an application thread: leave its device as it was)
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(device);
        (void)hipEventDestroy(ev);
        if (cur >= 0) (void)hipSetDevice(cur);
    }
};

struct ProcJob {            // one srcnn_process_u8 call; shared (read-only) by its per-context workers
    const unsigned char* rgb; unsigned w, h, d, dw, dh; int filter, cfilter, mode;
    unsigned char* out; unsigned char* conv;
    bool trace;
    AsyncLink* after = nullptr;     // asynchronous jobs: the previous job's link (kernels wait for it) ...
    AsyncLink* mine = nullptr;      // ... and this job's own (recorded behind its last kernel)
};

// Bands of a share [R0,R1) of a (dw x dh) output: about 10 / 30 / 30 / 20 / 7 / 3 % of the rows (a share of a multi-context call:
// 45 / 35 / 15 / 5) -- a short band first so that the GPU starts early, a very short one last because its D2H + fan-out cannot
// overlap anything --
// with every cut moved to where the band fills whole rounds of the persistent layer-1+2 grid (plan_cuts).  No band is larger
// than the workspace budget allows.
std::vector<unsigned> band_starts(unsigned R0, unsigned R1, unsigned dw, unsigned dh, bool first_share_of_many, int grid, int tile_rows)
{
    const unsigned cap = budget_band_rows(dw);
    // a short first band (the GPU starts after a twentieth of the input has been staged) and a very short last one (what is
    // left to copy and fan out after the last kernel).  Measured on 3840x2160 RGB x2, median of 22 calls, same box
    // (SRCNN_BANDS, tools/process_probe.py): 10/30/30/20/7/3 % 12.5-12.7 ms; 5/15/30/27/16/5/2 % 12.1-12.3 ms; a 3 % first
    // band or a 7-9 % second-to-last one are slower again (12.4 ms).
    // Below ~11 Mpx of output (1920x1080 x2: 3.1-3.3 ms either way, 3.3 with seven bands) the per-band costs outweigh the finer
    // head and tail, and the coarser plan stays; below 3 Mpx (only a small share of a multi-context call is banded at all) four
    // bands.  The plan follows the SHARE's size, lone or not: every context starts its own device after its own first band.
    static const double seven[] = {0.05, 0.15, 0.30, 0.27, 0.16, 0.05}, six[] = {0.10, 0.30, 0.30, 0.20, 0.07}, four[] = {0.45, 0.35, 0.15};
    const size_t share_px = (size_t)(R1 - R0) * dw;
    const bool large = share_px >= 11000000u;
    const bool tiny = share_px < 3000000u;       // a small share of a multi-context call (lone images this small are not banded)
    // SRCNN_BANDS="f0,f1,...": band fractions of a lone share for A/B runs (the last band is what is left)
    static const std::vector<double> env_plan = [] {
        std::vector<double> v;
        if (const char* e = settings().bands.empty() ? nullptr : settings().bands.c_str()) {
            double sum = 0;
            for (const char* q = e; *q;) {
                char* end = nullptr;
                const double f = strtod(q, &end);
                if (end == q) break;
                if (f > 0 && sum + f < 1.0 && v.size() < 15) { v.push_back(f); sum += f; }
                q = (*end == ',') ? end + 1 : end;
                if (*end != ',') break;
            }
        }
        return v;
    }();
    const bool custom = !env_plan.empty() && !first_share_of_many;
    const double* plan = custom ? env_plan.data() : (tiny ? four : (large ? seven : six));
    const int nplan = custom ? (int)env_plan.size() : (tiny ? 3 : (large ? 6 : 5));
    std::vector<unsigned> cuts = (R1 - R0 >= 512) ? plan_cuts(R0, R1, dw, dh, plan, nplan, grid, tile_rows) : std::vector<unsigned>{R0, R1};
    // enforce the budget: split anything larger than `cap` rows
    std::vector<unsigned> out{R0};
    for (size_t i = 1; i < cuts.size(); ++i) {
        unsigned a = out.back();
        const unsigned b = cuts[i];
        if (b <= a) continue;
        while (b - a > cap) { a += cap; out.push_back(a); }
        out.push_back(b);
    }
    return out;
}

// One context's share of a ProcessSRCNN call: output rows [R0,R1), pipelined over bands.
int process_share(Ctx& cx, const ProcJob& J, unsigned R0, unsigned R1, bool one_of_many)
{
    int rc = bind(cx);
    if (rc) return rc;
    TraceRange tr("srcnn_process_u8 ctx %d rows [%u,%u) of %ux%ux%u -> %ux%u", cx.index, R0, R1, J.w, J.h, J.d, J.dw, J.dh);
    if (J.after) J.after->wait_leased();
    LaneLease lease(cx);
    if (J.mine) J.mine->mark_leased();
    if (lease.rc) return lease.rc;
    ProcLane& L = *lease.lane;
    Workspace& ws = L.ws;
    hipStream_t s = L.st;
    std::vector<TableRef> tables;
    Call c;
    c.cx = &cx; c.s = s; c.ws = &ws; c.mode = J.mode; c.hold = &tables;
    const unsigned w = J.w, h = J.h, d = J.d, dw = J.dw, dh = J.dh;
    const size_t n = (size_t)w * h;
    const auto now = [] { return std::chrono::steady_clock::now(); };
    const auto t0 = now();

    // A lane that once served a much larger image gives its memory back before it grows for this one
    {
        const size_t need = n * d + (size_t)(R1 - R0) * dw * (d + 1 + 4 + 8) + (size_t)C2N * dw * std::min<size_t>(R1 - R0, budget_band_rows(dw)) * 4;
        const size_t have = ws.footprint() + L.pin_in_n + L.pin_out_n;
        if (have > (256u << 20) && have > 8 * need) { (void)wait_stream(L.st); (void)wait_stream(L.copy_st); L.release_buffers(); }
    }

    // ---- which source rows does this share read?  (the Y path's vertical taps + halo, and the chroma taps) ----
    unsigned lo = 0, hi = h;
    TableRef cv, ch_, yv, yh;
    const bool identity = (dw == w && dh == h);
    if (!identity) {
        if ((rc = y_path_source_rows(c, h, dh, J.filter, R0, R1, lo, hi))) return rc;
        if (dh != h) {
            if ((rc = get_table(c, J.cfilter, dh, h, cv))) return rc;
            unsigned clo, chi;
            cv->source_span(R0, R1, clo, chi);
            lo = std::min(lo, clo); hi = std::max(hi, std::min(chi, h));
        } else { lo = std::min(lo, R0); hi = std::max(hi, R1); }
    }
    // The fused shell reads the interleaved source directly (no split, no destination-size chroma planes); it needs an
    // up-scale in both axes with short monotone tables -- anything else takes the plane path below.
    // (the fused shell feeds the Y path an RGB source, which only k_rs2d can read: the switches that force the older plane
    //  resamplers therefore select the plane shell as well)
    bool fused_shell = !settings().shell_unfused && !settings().resample_2pass && !identity && dw > w && dh > h;
    if (fused_shell) {
        if ((rc = get_table(c, J.cfilter, dw, w, ch_))) return rc;
        if ((rc = get_table(c, J.filter, dh, h, yv))) return rc;
        if ((rc = get_table(c, J.filter, dw, w, yh))) return rc;
        // every band of this share must be acceptable to both fused kernels (Y from RGB, merge with on-the-fly chroma);
        // the Y path reads rows up to 6 beyond the share
        const unsigned ya = R0 >= 6 ? R0 - 6 : 0, yb = std::min(dh, R1 + 6);
        fused_shell = rs2d_fits(1, (int)w, (int)h, (int)dw, (int)dh, (int)ya, (int)yb, yv->view(), yh->view()) &&
                      rs2d_fits((int)d - 1, (int)w, (int)h, (int)dw, (int)dh, (int)R0, (int)R1, cv->view(), ch_->view());
    }

    // ---- device buffers (the lane's grow-only scratch) ----
    const size_t share_px = (size_t)(R1 - R0) * dw;
    //   bytes:  [source image (whole-frame geometry; only rows lo..hi are ever written/read)] [out bands] [conv bands]
    if ((rc = grow_ws(ws, ws.bytes, ws.bytes_n, n * d + share_px * d + share_px))) return rc;
    unsigned char* d_rgb = ws.bytes;
    unsigned char* d_out = ws.bytes + n * d;                 // row R0 at offset 0
    unsigned char* d_conv = d_out + share_px * d;
    //   planes: [Y' of the share] and, on the plane path only, [Y Cb Cr A at source size] [Cb' Cr' A' of the share]
    const size_t planes_need = share_px + (fused_shell ? 0 : 4 * n + 3 * share_px);
    if ((rc = grow_ws(ws, ws.planes, ws.planes_n, planes_need))) return rc;
    float* yp = ws.planes;                                    // Y', row R0 at offset 0
    float* sp[4] = {nullptr, nullptr, nullptr, nullptr};
    float* dp[4] = {yp, nullptr, nullptr, nullptr};
    if (!fused_shell) {
        for (int k = 0; k < 4; ++k) sp[k] = ws.planes + share_px + k * n;
        for (int k = 1; k < 4; ++k) dp[k] = ws.planes + share_px + 4 * n + (k - 1) * share_px;
    }

    // ---- bands ----
    int grid = 0, tile_rows = 0;
    if ((J.mode & 0xff) != SRCNN_MODE_FAST_F16) conv12_grid_info(cx.num_cus, &grid, &tile_rows);
    const std::vector<unsigned> cuts = band_starts(R0, R1, dw, dh, one_of_many, grid, tile_rows);
    const unsigned nb = (unsigned)cuts.size() - 1;
    unsigned max_band = 0;
    for (unsigned b = 0; b < nb; ++b) max_band = std::max(max_band, cuts[b + 1] - cuts[b]);
    const bool no_planes = c.mode == SRCNN_MODE_FAST_F16;           // the fused kernel has no layer-2 planes
    if (!no_planes && (rc = grow_ws(ws, ws.c2, ws.c2_n, (size_t)C2N * dw * std::min(dh, max_band + 4)))) return rc;
    if ((rc = grow_ws(ws, ws.up, ws.up_n, (size_t)dw * std::min(dh, max_band + 12)))) return rc;
    if (!fused_shell && (rc = grow_ws(ws, ws.tmp, ws.tmp_n, (size_t)std::max(w, dw) * std::max(h, std::min(dh, max_band + 12))))) return rc;
    while (L.band_events.size() < 3 * nb + 1) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, kBlockingEvent));
        L.band_events.push_back(e);
    }

    const size_t out_bytes = share_px * d;
    const bool small = out_bytes < (8u << 20) && !one_of_many;
    // Buffers the caller allocated page-locked with srcnn_host_alloc_pinned need no staging: the source rows go
    // to the device straight from the caller's image and every band lands straight in the caller's result.  That removes the
    // 125 MB of staging memcpy and the 100 MB fan-out per 4K image that bound a SEQUENCE of calls on the host
    // (profiles/r04_process_clock.txt: 10.8 ms per image blocking, 11.5 with two jobs in flight, for 9 ms of device work).
    const bool in_pinned = !small && pinned_by_library(J.rgb, n * d);
    const bool out_pinned = !small && pinned_by_library(J.out, (size_t)dw * dh * d) && (!J.conv || pinned_by_library(J.conv, (size_t)dw * dh));
    // small images: caller buffers the library page-locked itself are used in place, everything else goes through the lane's staging
    const bool small_in_locked = small && pinned_by_library(J.rgb, n * d);
    const bool small_out_locked = small && pinned_by_library(J.out, (size_t)dw * dh * d) && (!J.conv || pinned_by_library(J.conv, (size_t)dw * dh));
    // ---- stage-in.  Small images: the share's source rows in one go, straight from the caller's (pageable) buffer.  Large
    //      images: band by band -- stage_rows(upto) brings source rows [staged, upto) through the page-locked staging to the
    //      device (and, on the plane path, splits them), so the first band's kernels start after a third of the copy and the
    //      rest of it hides behind them. ----
    const size_t src_off = (size_t)lo * w * d, src_bytes = (size_t)(hi - lo) * w * d;
    unsigned staged = lo;                                      // source rows [lo, staged) are on the device
    unsigned n_staged = 0;                                     // slabs staged so far (one "rows in" event each, at most one per band)
    auto stage_rows = [&](unsigned upto) -> int {
        upto = std::min(std::max(upto, staged), hi);
        if (upto == staged) return SRCNN_OK;
        const size_t off = (size_t)staged * w * d, nbytes = (size_t)(upto - staged) * w * d;
        if (small) {
            // through the lane's own page-locked staging, asynchronously on the lane's stream (a pageable pointer never goes to a
            // HIP copy: HostBounce in srcnn_host.hpp; the call waits once, at its end)
            const unsigned char* from = J.rgb + off;
            if (!small_in_locked) {
                memcpy(L.pin_in + (off - src_off), J.rgb + off, nbytes);
                from = L.pin_in + (off - src_off);
            }
            HIP_TRY(hipMemcpyAsync(d_rgb + off, from, nbytes, hipMemcpyHostToDevice, s));
        } else {
            // on the lane's own input stream, so that the copy runs BESIDE the previous band's kernels.  The band's kernels may
            // not start before the rows are there: this thread waits for the copy (polling; the previous band keeps the device
            // busy meanwhile) and only then queues them.  A device-side hipStreamWaitEvent would do the same without the wait,
            // but on this runtime a cross-stream wait is resolved by a thread of the RUNTIME that stays busy from the call until
            // the event fires: 9.5 ms of CPU per 11.7 ms call, and no faster (tools/runtime_thread_probe.py,
            // profiles/r03_wait_cost.txt).  SRCNN_DEVICE_WAIT_IN=1 selects it for A/B runs.
            const unsigned char* from = J.rgb + off;
            if (!in_pinned) {
                parallel_memcpy(L.pin_in + (off - src_off), J.rgb + off, nbytes);
                from = L.pin_in + (off - src_off);
            }
            hipEvent_t ev = L.band_events[2 * nb + n_staged++];
            HIP_TRY(hipMemcpyAsync(d_rgb + off, from, nbytes, hipMemcpyHostToDevice, L.in_st));
            HIP_TRY(hipEventRecord(ev, L.in_st));
            if (settings().device_wait_in) HIP_TRY(hipStreamWaitEvent(s, ev, 0));
            else if (wait_event(ev) != hipSuccess) return fail(SRCNN_E_HIP, "stage-in copy");
        }
        if (!fused_shell)
            launch_rgb_split(d_rgb + off, (size_t)(upto - staged) * w, (int)d, sp[0] + (size_t)staged * w, sp[1] + (size_t)staged * w,
                             sp[2] + (size_t)staged * w, sp[3] + (size_t)staged * w, s);
        staged = upto;
        return SRCNN_OK;
    };
    auto band_source_end = [&](unsigned a, unsigned b, unsigned& upto) -> int {      // last source row (exclusive) band [a,b) reads
        upto = hi;
        if (identity) return SRCNN_OK;
        unsigned l2 = 0, h2 = hi;
        int r = y_path_source_rows(c, h, dh, J.filter, a, b, l2, h2);
        if (r) return r;
        if (cv) { unsigned cl, chh; cv->source_span(a, b, cl, chh); h2 = std::max(h2, std::min(chh, h)); }
        else h2 = std::max(h2, std::min(b, h));
        upto = std::min(h2, hi);
        return SRCNN_OK;
    };
    std::thread prefault;
    bool prefaulting = false;
    if (small) {
        if (!small_in_locked && (rc = grow_pinned(cx, L.pin_in, L.pin_in_n, src_bytes))) return rc;
        if (!small_out_locked && (rc = grow_pinned(cx, L.pin_out, L.pin_out_n, out_bytes + share_px))) return rc;
    }
    if (!small) {
        if (!in_pinned && (rc = grow_pinned(cx, L.pin_in, L.pin_in_n, src_bytes))) return rc;
        if (!out_pinned && (rc = grow_pinned(cx, L.pin_out, L.pin_out_n, out_bytes + share_px))) return rc;
        unsigned char* o0 = J.out + (size_t)R0 * dw * d;
        unsigned char* c0 = J.conv ? J.conv + (size_t)R0 * dw : nullptr;
        if (!out_pinned) {
            hint_huge_pages(o0, out_bytes);
            if (c0) hint_huge_pages(c0, share_px);
        }
        const int pf_threads = (int)settings().prefault_threads;
        if (settings().prefault && !out_pinned) prefaulting = try_thread(prefault, [=] { prefault_pages(o0, out_bytes, pf_threads); if (c0) prefault_pages(c0, share_px); });
    }
    struct JoinPrefault {                                      // whatever path leaves this function: the helper is joined first
        std::thread& t; bool& on;
        ~JoinPrefault() { if (on) t.join(); }
    } join_prefault{prefault, prefaulting};
    const auto t1 = now();

    // asynchronous chain (see AsyncLink): called once, right before this job's first launch on the compute stream
    bool chained_in = false;
    auto chain_in = [&]() -> int {
        if (chained_in || !J.after) return SRCNN_OK;
        chained_in = true;
        AsyncLink& a = *J.after;
        {
            std::unique_lock<std::mutex> lk(a.m);
            a.cv.wait(lk, [&] { return a.done; });
        }
        // (settled links never change again: valid / ev are read without the lock, which the wait below must not hold)
        if (a.valid && a.ev) {
            // resolved on the HOST by default (this thread polls the predecessor's last-kernel event, then queues): a device-side
            // hipStreamWaitEvent across streams is resolved by a thread of the runtime on this ROCm and measured slower
            // (SRCNN_ASYNC_CHAIN=2 selects it for A/B runs)
            if (settings().async_chain == 2) HIP_TRY(hipStreamWaitEvent(s, a.ev, 0));
            else if (wait_event(a.ev) != hipSuccess) return fail(SRCNN_E_HIP, "waiting for the previous asynchronous job's kernels");
        }
        return SRCNN_OK;
    };
    auto chain_out = [&]() {          // right after this job's last launch on the compute stream
        if (!J.mine) return;
        AsyncLink& m = *J.mine;
        m.device = cx.device;
        bool ok = m.ev || hipEventCreateWithFlags(&m.ev, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventRecord(m.ev, s) == hipSuccess;
        m.settle(ok);
    };

    const YSource ysrc = fused_shell ? YSource::from_rgb(d_rgb, (int)d) : YSource::from_plane(sp[0]);
    auto run_band = [&](unsigned a, unsigned b) -> int {       // kernels of output rows [a,b)
        const size_t p0 = (size_t)(a - R0) * dw, pn = (size_t)(b - a) * dw;
        int r = y_path_rows(c, ysrc, w, h, dw, dh, J.filter, a, b, yp + p0);
        if (r) return r;
        if (fused_shell) {
            const DevAxisTable tv = cv->view(), th = ch_->view();
            if (!launch_merge_fused(d_rgb, (int)w, (int)h, (int)d, yp + p0, d_out + p0 * d, J.conv ? d_conv + p0 : nullptr,
                                    (int)dw, (int)dh, (int)a, (int)(b - a), tv, th, s))
                return fail(SRCNN_E_UNSUPPORTED, "fused colour shell refused a shape it was selected for");
        } else {
            for (unsigned k = 1; k < d; ++k)
                if ((r = resample_rows_range(c, sp[k], w, h, dw, dh, J.cfilter, a, b, dp[k] + p0))) return r;
            launch_ycc_merge(yp + p0, dp[1] + p0, dp[2] + p0, dp[3] ? dp[3] + p0 : nullptr, pn, (int)d, d_out + p0 * d,
                             J.conv ? d_conv + p0 : nullptr, s);
        }
        HIP_TRY(hipGetLastError());
        return SRCNN_OK;
    };

    if (small) {
        // small image: one shot on the lane's stream
        if ((rc = chain_in())) return rc;
        if ((rc = stage_rows(hi))) return rc;
        if ((rc = run_band(R0, R1))) return rc;
        chain_out();
        unsigned char* to_rgb = small_out_locked ? J.out + (size_t)R0 * dw * d : L.pin_out;
        unsigned char* to_conv = small_out_locked ? (J.conv ? J.conv + (size_t)R0 * dw : nullptr) : L.pin_out + out_bytes;
        HIP_TRY(hipMemcpyAsync(to_rgb, d_out, out_bytes, hipMemcpyDeviceToHost, s));
        if (J.conv) HIP_TRY(hipMemcpyAsync(to_conv, d_conv, share_px, hipMemcpyDeviceToHost, s));
        HIP_TRY(wait_stream(s));
        if (!small_out_locked) {
            memcpy(J.out + (size_t)R0 * dw * d, L.pin_out, out_bytes);
            if (J.conv) memcpy(J.conv + (size_t)R0 * dw, L.pin_out + out_bytes, share_px);
        }
        return SRCNN_OK;
    }

    // Large image: the reference's only benchmark is the wall time of this call (src/test.cpp:653-672), and for a GPU
    // that is dominated by moving ~4 B per output pixel to and from pageable host memory.  So: page-locked staging on both
    // sides, the output produced in bands (bit-identical to the whole frame, tests/test_gpu_parity.py::
    // test_bands_equal_whole_frame), each band's D2H on the copy stream while the next band computes, and a helper thread
    // that fans each landed band out to the caller's buffers.  The helper blocks on events; it never spins.
    unsigned char* pin_rgb = out_pinned ? nullptr : L.pin_out;
    unsigned char* pin_conv = out_pinned ? nullptr : L.pin_out + out_bytes;
    std::atomic<int> copy_err{0};
    // SRCNN_TRACE stamps, microseconds since entry: first band queued, last band's kernels done, last band landed in staging
    std::atomic<long> us_first_queued{0}, us_kernels_done{0}, us_landed{0};
    const auto since = [&] { return (long)std::chrono::duration_cast<std::chrono::microseconds>(now() - t0).count(); };
    Handoff enqueued;                       // bands whose kernels have been queued (their "computed" event recorded)
    auto d2h_band = [&](unsigned b) {
        const size_t p0 = (size_t)(cuts[b] - R0) * dw, pn = (size_t)(cuts[b + 1] - cuts[b]) * dw;
        const hipError_t kd = wait_event(L.band_events[2 * b]);
        if (b + 1 == nb) us_kernels_done = since();
        const size_t g0 = (size_t)cuts[b] * dw;
        unsigned char* to_rgb = out_pinned ? J.out + g0 * d : pin_rgb + p0 * d;          // page-locked result: no staging
        unsigned char* to_conv = out_pinned ? (J.conv ? J.conv + g0 : nullptr) : pin_conv + p0;
        if (kd != hipSuccess ||
            hipMemcpyAsync(to_rgb, d_out + p0 * d, pn * d, hipMemcpyDeviceToHost, L.copy_st) != hipSuccess ||
            (J.conv && hipMemcpyAsync(to_conv, d_conv + p0, pn, hipMemcpyDeviceToHost, L.copy_st) != hipSuccess) ||
            hipEventRecord(L.band_events[2 * b + 1], L.copy_st) != hipSuccess) { copy_err = 1; return false; }
        return true;
    };
    auto fan_band = [&](unsigned b) {
        TraceRange tf("srcnn fan-out band %u", b);
        if (wait_event(L.band_events[2 * b + 1]) != hipSuccess) { copy_err = 1; return; }
        if (b + 1 == nb) us_landed = since();
        if (out_pinned) return;                                 // the band landed in the caller's buffer itself
        const size_t p0 = (size_t)(cuts[b] - R0) * dw, pn = (size_t)(cuts[b + 1] - cuts[b]) * dw;
        const size_t g0 = (size_t)cuts[b] * dw;
        parallel_memcpy(J.out + g0 * d, pin_rgb + p0 * d, pn * d);
        if (J.conv) parallel_memcpy(J.conv + g0, pin_conv + p0, pn);
    };
    // Two helpers, so that neither kind of waiting delays the other: the COPIER waits for a band's kernels and queues its D2H
    // on the idle copy stream at once; the FANNER waits for a landed band and copies it out to the caller's buffers.  (One
    // helper doing both fanned band b-1 out only after band b's kernels had finished -- a marker trace showed the first
    // fan-out starting 8 ms into a 13.9 ms call and the last two sitting in the tail.)  Both sleep or poll; neither spins.
    Handoff landed_q;                       // bands whose D2H has been queued (their "landed" event recorded)
    std::thread copier, fanner;
    const bool threaded = try_thread(copier, [&] {
        (void)hipSetDevice(cx.device);
        for (unsigned b = 0; b < nb; ++b) {
            if (!enqueued.wait_for(b) || !d2h_band(b)) { landed_q.cancel(); return; }
            landed_q.publish(b + 1);
        }
    });
    const bool fan_threaded = threaded && try_thread(fanner, [&] {
        (void)hipSetDevice(cx.device);
        for (unsigned b = 0; b < nb; ++b) {
            if (!landed_q.wait_for(b)) return;
            fan_band(b);
        }
    });
    int launch_rc = SRCNN_OK;
    for (unsigned b = 0; b < nb; ++b) {
        TraceRange tb("srcnn band %u [%u,%u)", b, cuts[b], cuts[b + 1]);
        unsigned upto = hi;
        launch_rc = band_source_end(cuts[b], cuts[b + 1], upto);
        // the plane shell splits the staged rows with a kernel on the compute stream: chain before it; the fused shell's first
        // launch is the band's resampler: stage first (the copy runs beside the previous job's kernels), then chain
        if (!launch_rc && !fused_shell) launch_rc = chain_in();
        if (!launch_rc) launch_rc = stage_rows(b + 1 == nb ? hi : upto);
        if (!launch_rc) launch_rc = chain_in();
        if (!launch_rc) launch_rc = run_band(cuts[b], cuts[b + 1]);
        if (!launch_rc && hipEventRecord(L.band_events[2 * b], s) != hipSuccess) launch_rc = fail(SRCNN_E_HIP, "band %u event record failed", b);
        if (!launch_rc && b + 1 == nb) chain_out();
        if (launch_rc) { enqueued.cancel(); break; }
        if (b == 0) us_first_queued = since();
        if (threaded) enqueued.publish(b + 1);
        else if (d2h_band(b)) fan_band(b);
    }
    if (threaded) copier.join();
    if (fan_threaded) fanner.join();
    else if (threaded && !launch_rc && !copy_err)
        for (unsigned b = 0; b < nb; ++b) fan_band(b);            // the second helper could not be started: fan out here
    const auto t2 = now();
    // everything this call queued has completed by now (the helper waited for the last D2H event); these return at once
    hipError_t e1 = wait_stream(s), e2 = wait_stream(L.copy_st);
    if (launch_rc) return launch_rc;
    if (e1 != hipSuccess || e2 != hipSuccess || copy_err) return fail(SRCNN_E_HIP, "pipeline failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    HIP_TRY(hipGetLastError());
    {
        PhaseRecord& P = g_phases;
        P.v[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();       // setup: lease, tables, scratch, staging, helper threads
        P.v[1] = us_first_queued.load() * 1e-3;                                     // first band staged in and its kernels queued
        P.v[2] = us_kernels_done.load() * 1e-3;                                     // last band's kernels finished on the device
        P.v[3] = us_landed.load() * 1e-3;                                           // last band's D2H landed (staging or caller's pinned buffer)
        P.v[4] = std::chrono::duration<double, std::milli>(t2 - t0).count();       // last band fanned out to the caller's buffer
        P.v[5] = (double)nb;
        P.n = 6;
    }
    if (J.trace) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        };
        fprintf(stderr, "srcnn_process_u8 ctx %d (device %d) rows [%u,%u) of %ux%ux%u x%.2f: setup %.2f ms, %u bands (compute || D2H || fan-out) %.2f ms%s; "
                        "since entry: first band queued %.2f, last kernels done %.2f, last band landed %.2f, fanned out %.2f ms\n",
                cx.index, cx.device, R0, R1, w, h, d, (double)dw / w, ms(t0, t1), nb, ms(t1, t2), fused_shell ? ", fused shell" : "",
                us_first_queued.load() * 1e-3, us_kernels_done.load() * 1e-3, us_landed.load() * 1e-3, ms(t0, t2));
    }
    return SRCNN_OK;
}

}  // namespace
}  // namespace srcnn

using namespace srcnn;

extern "C" {

// Diagnostic: how the process's last srcnn_y_upscale2x_f32_stream call launched its frames (summed over the contexts): frames
// replayed from a hipGraph, frames launched plainly (the first frame of a slot always is: capture needs one eager run), and
// whether use_graph = 1 gave the graphs up because replay burnt host CPU (SRCNN_GRAPH_MAX_CPU_PCT).
int srcnn_debug_stream_mode(unsigned* graph_frames, unsigned* plain_frames, int* fell_back)
{
    if (graph_frames) *graph_frames = g_stream_graph_frames.load();
    if (plain_frames) *plain_frames = g_stream_plain_frames.load();
    if (fell_back) *fell_back = g_stream_fell_back.load();
    return SRCNN_OK;
}

// Diagnostic: the phase stamps of the calling thread's last large (banded) srcnn_process_u8 / ProcessSRCNN share, in
// milliseconds since the share began: setup done, first band queued, last kernels done, last band landed, fanned out, and the
// band count.  Returns how many values exist (0: no banded call ran on this thread), writes at most `cap`.
int srcnn_debug_process_phases(double* ms, int cap)
{
    const PhaseRecord& P = g_phases;
    for (int i = 0; i < P.n && i < cap; ++i) ms[i] = P.v[i];
    return P.n;
}

// Test hook (no device needed): the band cut points srcnn_process_u8 would use for output rows [r0, r1) of a dw-wide image
// under the current workspace limit.  Writes up to `cap` cut points (first = r0, last = r1), returns how many there are.
int srcnn_debug_band_plan(unsigned r0, unsigned r1, unsigned dw, int one_of_many, unsigned* cuts, int cap)
{
    if (r1 <= r0 || dw == 0) return fail(SRCNN_E_ARG, "empty row range");
    int grid = 0, tile_rows = 0;
    conv12_grid_info(256, &grid, &tile_rows);          // an MI355X: 256 CUs
    const std::vector<unsigned> c = band_starts(r0, r1, dw, r1, one_of_many != 0, grid, tile_rows);
    for (int i = 0; i < (int)c.size() && i < cap; ++i) cuts[i] = c[i];
    return (int)c.size();
}

// ---- host-pointer conveniences -----------------------------------------------------------------
int srcnn_y_path_f32(const float* in, unsigned w, unsigned h, unsigned dw, unsigned dh, int filter, float* out)
{
    int rc;
    if ((rc = check_plane(in, w, h, out))) return rc;
    if (dw == 0 || dh == 0) return fail(SRCNN_E_SCALE, "scaled size %ux%u", dw, dh);
    if (!cur_ctx()) return SRCNN_E_NODEVICE;
    // Device buffers kept per context (grow-only; srcnn_trim gives them back), copies through copy_*_any: neither a
    // hipMalloc / hipFree pair per call (a device-wide sync each) nor a pageable pointer handed to the runtime (HostBounce).
    Ctx& cx = *cur_ctx();
    if ((rc = bind(cx))) return rc;
    const size_t in_n = (size_t)w * h, out_n = (size_t)dw * dh;
    HostCallBuffers& hb = cx.host_call;
    std::lock_guard<std::mutex> one(hb.mu);
    if ((rc = grow(hb.d_in, hb.d_in_n, in_n))) return rc;
    if ((rc = grow(hb.d_out, hb.d_out_n, out_n))) return rc;
    if ((rc = copy_h2d_any(cx, hb.d_in, in, sizeof(float) * in_n, nullptr))) return rc;
    if ((rc = srcnn_y_path_f32_dev(hb.d_in, w, h, dw, dh, filter, hb.d_out, nullptr))) return rc;
    return copy_d2h_any(cx, out, hb.d_out, sizeof(float) * out_n, nullptr);       // same (default) stream: ordered behind the kernels
}

int srcnn_y_upscale2x_f32(const float* in, unsigned w, unsigned h, float* out)
{
    return srcnn_y_path_f32(in, w, h, 2 * w, 2 * h, SRCNN_FILTER_BICUBIC, out);
}

int srcnn_y_upscale2x_f32_stream(const float* in, unsigned w, unsigned h, unsigned nframes, float* out, int use_graph)
{
    int rc;
    if ((rc = check_plane(in, w, h, out))) return rc;
    if (nframes == 0) return fail(SRCNN_E_ARG, "nframes == 0");
    Ctx* cur = cur_ctx();
    if (!cur) return SRCNN_E_NODEVICE;
    const size_t in_n = (size_t)w * h, out_n = in_n * 4;
    const size_t in_b = in_n * sizeof(float), out_b = out_n * sizeof(float);
    const int mode = G.mode.load();
    g_stream_graph_frames = 0; g_stream_plain_frames = 0; g_stream_fell_back = 0;

    // page-lock the caller's frames so the copies are truly asynchronous -- unless they already are (buffers from
    // srcnn_host_alloc_pinned / hipHostMalloc: registering a gigabyte again costs milliseconds per call); harmless if it fails
    const bool pin_in = is_pinned(in), pin_out = is_pinned(out);
    const bool reg_in = !pin_in && hipHostRegister(const_cast<float*>(in), in_b * nframes, hipHostRegisterPortable) == hipSuccess;
    const bool reg_out = !pin_out && hipHostRegister(out, out_b * nframes, hipHostRegisterPortable) == hipSuccess;
    (void)hipGetLastError();
    const bool in_locked = pin_in || reg_in, out_locked = pin_out || reg_out;
    if (settings().trace)
        fprintf(stderr, "srcnn_y_upscale2x_f32_stream %ux%u x %u frames, use_graph %d: source %s, result %s\n", w, h, nframes, use_graph,
                pin_in ? "page-locked" : reg_in ? "page-locked for this call" : "pageable (bounced)",
                pin_out ? "page-locked" : reg_out ? "page-locked for this call" : "pageable (bounced)");

    // Frames are independent: with several contexts they are dealt out in contiguous chunks, one worker per context
    // (no data-path exchange at all; each context runs its own two-slot pipeline).
    const unsigned nctx = (unsigned)std::min<unsigned>((unsigned)context_count(), nframes);
    if (nctx <= 1) {
        rc = stream_on_ctx(*cur, in, w, h, nframes, out, use_graph, mode, in_locked, out_locked);
    } else {
        std::vector<int> rcs(nctx, SRCNN_OK);
        std::vector<std::string> errs(nctx);
        std::vector<std::thread> th(nctx);
        auto work = [&](unsigned k) {
            const unsigned f0 = (unsigned)((unsigned long long)nframes * k / nctx), f1 = (unsigned)((unsigned long long)nframes * (k + 1) / nctx);
            Ctx* cx = context_at((int)k);
            rcs[k] = cx ? stream_on_ctx(*cx, in + f0 * in_n, w, h, f1 - f0, out + f0 * out_n, use_graph, mode, in_locked, out_locked) : SRCNN_E_NODEVICE;
            if (rcs[k]) errs[k] = srcnn_last_error();
        };
        std::vector<bool> started(nctx, false);
        for (unsigned k = 1; k < nctx; ++k) started[k] = try_thread(th[k], [&, k] { work(k); });
        work(0);
        for (unsigned k = 1; k < nctx; ++k) { if (started[k]) th[k].join(); else work(k); }
        for (unsigned k = 0; k < nctx; ++k)
            if (rcs[k]) { rc = rcs[k]; set_last_error(errs[k].c_str()); break; }
        (void)bind(*cur);
    }
    if (reg_in) (void)hipHostUnregister(const_cast<float*>(in));
    if (reg_out) (void)hipHostUnregister(out);
    return rc;
}

int srcnn_y_upscale2x_f32_batch(const float* in, unsigned w, unsigned h, unsigned nframes, float* out)
{
    return srcnn_y_upscale2x_f32_stream(in, w, h, nframes, out, 0);
}

namespace {
int process_u8_impl(const unsigned char* rgb, unsigned w, unsigned h, unsigned d, float multiply, int filter,
                    unsigned char* out, unsigned char* conv_opt, int mode, AsyncLink* after, AsyncLink* mine);
}

int srcnn_process_u8(const unsigned char* rgb, unsigned w, unsigned h, unsigned d, float multiply, int filter,
                     unsigned char* out, unsigned char* conv_opt)
{
    return process_u8_impl(rgb, w, h, d, multiply, filter, out, conv_opt, G.mode.load(), nullptr, nullptr);
}

namespace {
int process_u8_impl(const unsigned char* rgb, unsigned w, unsigned h, unsigned d, float multiply, int filter,
                    unsigned char* out, unsigned char* conv_opt, int mode, AsyncLink* after, AsyncLink* mine)
{
    if (!rgb || !out || w == 0 || h == 0 || d == 0) return fail(SRCNN_E_ARG, "NULL pointer or zero dimension");
    if (d != 3 && d != 4) return fail(SRCNN_E_UNSUPPORTED, "depth %u: the reference reads uninitialised planes for d<3 (src/libsrcnn.cpp:235-236)", d);
    if ((float)w * multiply <= 0.f || (float)h * multiply <= 0.f) return fail(SRCNN_E_SCALE, "non-positive scaled size");
    if (filter < 0 || filter > 4) return fail(SRCNN_E_ARG, "bad filter %d", filter);
    Ctx* cur = cur_ctx();
    if (!cur) return SRCNN_E_NODEVICE;
    const unsigned dw = (unsigned)((float)w * multiply), dh = (unsigned)((float)h * multiply);   // src/libsrcnn.cpp:662-663
    if (dw == 0 || dh == 0) return fail(SRCNN_E_SCALE, "scaled size %ux%u", dw, dh);
    if ((unsigned long long)w * h > 0x7fffffffULL || (unsigned long long)dw * dh > 0x7fffffffULL)
        return fail(SRCNN_E_UNSUPPORTED, "plane too large");
    ProcJob J;
    J.rgb = rgb; J.w = w; J.h = h; J.d = d; J.dw = dw; J.dh = dh; J.filter = filter;
    J.cfilter = (filter == SRCNN_FILTER_NEAREST) ? SRCNN_FILTER_NEAREST : SRCNN_FILTER_BILINEAR;   // src/libsrcnn.cpp:701-713
    J.mode = mode; J.out = out; J.conv = conv_opt;
    J.trace = settings().trace;

    // Everything runs on lanes leased for this call only (see ProcLane): concurrent ProcessSRCNN calls from several host
    // threads are independent, like the reference's.  A large image is dealt over all contexts of the process in
    // contiguous row shares proportional to nothing but their count (the devices of a node are identical): the
    // reference's one call saturates its whole machine (src/libsrcnn.cpp:665,791-798,817-824), and so does this one.
    const size_t out_bytes = (size_t)dw * dh * d;
    unsigned shares = 1;
    if (out_bytes >= (8u << 20)) shares = std::max(1u, std::min<unsigned>((unsigned)context_count(), dh / 256u));
    if (shares <= 1) {
        J.after = after; J.mine = mine;            // the chain orders kernels on ONE device; a dealt-out image runs unchained
        return process_share(*cur, J, 0, dh, false);
    }
    if (mine) mine->settle(false);                 // nothing to wait for: the next job starts beside this one

    std::vector<int> rcs(shares, SRCNN_OK);
    std::vector<std::string> errs(shares);
    std::vector<std::thread> th(shares);
    // the calling thread's context takes share 0, the others follow in context order
    auto ctx_of_share = [&](unsigned k) { return context_at((int)((cur->index + k) % (unsigned)context_count())); };
    auto work = [&](unsigned k) {
        const unsigned r0 = (unsigned)((unsigned long long)dh * k / shares) & ~15u;
        const unsigned r1 = (k + 1 == shares) ? dh : ((unsigned)((unsigned long long)dh * (k + 1) / shares) & ~15u);
        Ctx* cx = ctx_of_share(k);
        rcs[k] = cx ? process_share(*cx, J, r0, r1, true) : SRCNN_E_NODEVICE;
        if (rcs[k]) errs[k] = srcnn_last_error();
    };
    std::vector<bool> started(shares, false);
    for (unsigned k = 1; k < shares; ++k) started[k] = try_thread(th[k], [&, k] { work(k); });
    work(0);
    for (unsigned k = 1; k < shares; ++k) { if (started[k]) th[k].join(); else work(k); }
    (void)bind(*cur);
    for (unsigned k = 0; k < shares; ++k)
        if (rcs[k]) { set_last_error(errs[k].c_str()); return rcs[k]; }
    return SRCNN_OK;
}
}  // namespace

// ---- the same call, asynchronous: begin() returns at once, wait() joins ----
// A caller that upscales a SEQUENCE of images keeps the device busy across calls this way: while image i's last band is
// copied out and fanned into its result buffer (the tail of the call, ~1 ms at 4K during which a synchronous caller's GPU
// idles, and after which the same kernels run 9-15 % slower for a while: DESIGN.md 4.5), image i+1's stage-in and first
// bands are already queued on another lane.  Each job runs srcnn_process_u8 on a thread of its own with the caller's
// current context; up to SRCNN_MAX_LANES jobs per context make progress at once, further ones queue for a lane like
// concurrent synchronous callers do.  The buffers must stay valid and untouched until wait() returns.
// The KERNELS of consecutive jobs of a context are chained (AsyncLink): job k+1's first kernel waits on the device for job k's
// last, while its stage-in copy and job k's copy-out run beside them -- two jobs sharing the device measured SLOWER than
// blocking calls (13.2 vs 10.7 ms per 4K image, profiles/r04_process_clock.txt).
}  // extern "C"

namespace {
struct AsyncJob {
    std::thread th;
    int rc = SRCNN_OK;
    std::string err;
    std::shared_ptr<AsyncLink> after, mine;
};
std::mutex g_async_mu;
// per context: the link of the job begun last.  On the heap and never destroyed by a static destructor (the HIP runtime may be
// gone by then, and a link owns an event); srcnn_shutdown drops the links while the devices are still there.
std::map<int, std::shared_ptr<AsyncLink>>& async_tails()
{
    static auto* m = new std::map<int, std::shared_ptr<AsyncLink>>;
    return *m;
}
}  // namespace

namespace srcnn {
void async_chain_reset()
{
    std::lock_guard<std::mutex> lk(g_async_mu);
    async_tails().clear();
}
}  // namespace srcnn

extern "C" {

int srcnn_process_u8_begin(const unsigned char* rgb, unsigned w, unsigned h, unsigned d, float multiply, int filter,
                           unsigned char* out, unsigned char* conv_opt, void** job)
{
    if (!job) return fail(SRCNN_E_ARG, "job == NULL");
    *job = nullptr;
    Ctx* cur = cur_ctx();
    if (!cur) return SRCNN_E_NODEVICE;
    const int ctx_index = cur->index;
    AsyncJob* a = new (std::nothrow) AsyncJob;
    if (!a) return fail(SRCNN_E_OUTALLOC, "out of memory");
    // the numerics mode of the image is the one in force NOW, not whenever the worker thread gets going: srcnn_amd.h promises
    // that srcnn_set_mode / srcnn_set_relaxation only affect calls that start later
    const int mode = G.mode.load();
    if (settings().async_chain != 0) {
        a->mine = std::make_shared<AsyncLink>();
        std::lock_guard<std::mutex> lk(g_async_mu);
        auto& tail = async_tails()[ctx_index];
        a->after = tail;
        tail = a->mine;
    }
    auto work = [=] {
        (void)srcnn_set_context(ctx_index);                   // the worker inherits the caller's current context
        a->rc = process_u8_impl(rgb, w, h, d, multiply, filter, out, conv_opt, mode, a->after.get(), a->mine.get());
        if (a->rc) a->err = srcnn_last_error();
        if (a->mine) a->mine->settle(false);                  // a job that never recorded its link (error, dealt-out image) releases its successor
    };
    if (!try_thread(a->th, work)) work();                     // no thread to be had: degrade to the synchronous call
    *job = a;
    return SRCNN_OK;
}

int srcnn_process_u8_wait(void* job)
{
    if (!job) return fail(SRCNN_E_ARG, "job == NULL");
    AsyncJob* a = static_cast<AsyncJob*>(job);
    if (a->th.joinable()) a->th.join();
    const int rc = a->rc;
    if (rc) set_last_error(a->err.c_str());
    delete a;
    return rc;
}

// ------------------------------------------------------------------------------------------------------------------
// ONE device-resident frame tiled over all contexts of this process (BASELINE config #4 from a single process).
// d_in / d_out live on the calling thread's current context (the root).
// ------------------------------------------------------------------------------------------------------------------
int srcnn_y_upscale2x_f32_node_dev(const float* d_in, unsigned w, unsigned h, float* d_out, int sub_bands)
{
    int rc;
    if ((rc = check_plane(d_in, w, h, d_out))) return rc;
    Ctx* root = cur_ctx();
    if (!root) return SRCNN_E_NODEVICE;
    const unsigned dw = 2 * w, dh = 2 * h;
    if (dh > (1u << 20) || dw > 0x7fffffu) return fail(SRCNN_E_UNSUPPORTED, "output %ux%u too large", dw, dh);
    const unsigned nctx = std::max(1u, std::min<unsigned>((unsigned)context_count(), dh / 64u));
    const unsigned nsub = (unsigned)std::max(1, std::min(sub_bands <= 0 ? 4 : sub_bands, 16));
    const int mode = G.mode.load();
    // the caller's earlier work on the root's default stream (uploads, a previous frame) must be visible to the workers
    HIP_TRY(hipDeviceSynchronize());

    std::vector<int> rcs(nctx, SRCNN_OK);
    std::vector<std::string> errs(nctx);
    auto work = [&](unsigned k) -> int {
        Ctx& cx = *context_at((int)((root->index + k) % (unsigned)context_count()));
        int r = bind(cx);
        if (r) return r;
        std::lock_guard<std::mutex> nlk(cx.node_mu);
        NodeLane& N = cx.node;
        if (!N.st) HIP_TRY(hipStreamCreateWithFlags(&N.st, hipStreamNonBlocking));
        if (!N.copy_st) HIP_TRY(hipStreamCreateWithFlags(&N.copy_st, hipStreamNonBlocking));
        unsigned R0 = 0, Rn = 0;
        (void)srcnn_band_rows(dh, (int)k, (int)nctx, &R0, &Rn);          // the same partition the tiled pieces are planned on
        const unsigned R1 = R0 + Rn;
        if (R1 <= R0) return SRCNN_OK;
        std::vector<TableRef> tables;
        Call c;
        c.cx = &cx; c.s = N.st; c.ws = &N.ws; c.mode = mode; c.hold = &tables;
        const bool is_root = (&cx == root);
        // source rows of this band, pulled from the root device into a buffer with whole-frame geometry
        unsigned lo = 0, hi = h;
        if ((r = y_path_source_rows(c, h, dh, SRCNN_FILTER_BICUBIC, R0, R1, lo, hi))) return r;
        const float* src = d_in;
        if (!is_root) {
            if ((r = grow(N.in, N.in_n, (size_t)w * h))) return r;
            if ((r = grow(N.band, N.band_n, (size_t)(R1 - R0) * dw))) return r;
            HIP_TRY(hipMemcpyPeerAsync(N.in + (size_t)lo * w, cx.device, d_in + (size_t)lo * w, root->device,
                                       sizeof(float) * (size_t)(hi - lo) * w, N.st));
            src = N.in;
        }
        while (N.events.size() < nsub + 2) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, kBlockingEvent));
            N.events.push_back(e);
        }
        // sub-bands: the kernels of sub-band i+1 are queued before the host waits for sub-band i and pushes it to the root
        // on the copy stream, so the push (one xGMI link per peer, all peers concurrently) hides behind compute
        // pieces: large first, short last, each cut where it fills whole rounds of the layer-1+2 grid (tiled_cuts); fewer than
        // nsub pieces come back for a short band, the rest are empty
        std::vector<unsigned> cut = tiled_cuts(dw, dh, (int)k, (int)nctx, (int)nsub);
        while (cut.size() < nsub + 1) cut.push_back(R1);
        auto launch = [&](unsigned i) -> int {
            if (cut[i + 1] <= cut[i]) return SRCNN_OK;
            float* dst = is_root ? d_out + (size_t)cut[i] * dw : N.band + (size_t)(cut[i] - R0) * dw;
            int q = y_path_range(c, src, w, h, dw, dh, SRCNN_FILTER_BICUBIC, cut[i], cut[i + 1], dst);
            if (q) return q;
            HIP_TRY(hipEventRecord(N.events[i], N.st));
            return SRCNN_OK;
        };
        auto push = [&](unsigned i) -> int {
            if (is_root || cut[i + 1] <= cut[i]) return SRCNN_OK;
            HIP_TRY(wait_event(N.events[i]));
            HIP_TRY(hipMemcpyPeerAsync(d_out + (size_t)cut[i] * dw, root->device, N.band + (size_t)(cut[i] - R0) * dw, cx.device,
                                       sizeof(float) * (size_t)(cut[i + 1] - cut[i]) * dw, N.copy_st));
            return SRCNN_OK;
        };
        if ((r = launch(0))) return r;
        for (unsigned i = 0; i < nsub; ++i) {
            if (i + 1 < nsub && (r = launch(i + 1))) return r;
            if ((r = push(i))) return r;
        }
        // wait for both queues by polling their events (8 workers inside hipStreamSynchronize would hold 8 host cores)
        HIP_TRY(hipEventRecord(N.events[nsub], N.st));
        HIP_TRY(hipEventRecord(N.events[nsub + 1], N.copy_st));
        HIP_TRY(wait_event(N.events[nsub]));
        HIP_TRY(wait_event(N.events[nsub + 1]));
        return SRCNN_OK;
    };
    std::vector<std::thread> th(nctx);
    std::vector<bool> started(nctx, false);
    auto run = [&](unsigned k) { rcs[k] = work(k); if (rcs[k]) errs[k] = srcnn_last_error(); };
    for (unsigned k = 1; k < nctx; ++k) started[k] = try_thread(th[k], [&, k] { run(k); });
    run(0);
    for (unsigned k = 1; k < nctx; ++k) { if (started[k]) th[k].join(); else run(k); }
    (void)bind(*root);
    for (unsigned k = 0; k < nctx; ++k)
        if (rcs[k]) { set_last_error(errs[k].c_str()); return rcs[k]; }
    return SRCNN_OK;
}

}  // extern "C"
