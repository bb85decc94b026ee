// wait_cost.hip -- what does a host thread burn while it waits for the GPU?  A ~5 ms kernel, then one of the host waits;
// thread CPU time (CLOCK_THREAD_CPUTIME_ID) and process CPU time (all threads, incl. the runtime's own) per wait.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <ctime>
#include <thread>
#include <chrono>

__global__ void spin(long long cycles, int* sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 1000) *sink = 1;
}

static double now_thread() { timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double now_proc() { timespec t; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static double now_wall() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char** argv)
{
    const int flags_sel = argc > 1 ? atoi(argv[1]) : 0;      // 0 default, 1 blocking-sync device flag, 2 yield
    if (flags_sel == 1) hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    if (flags_sel == 2) hipSetDeviceFlags(hipDeviceScheduleYield);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e_spin, e_block;
    hipEventCreateWithFlags(&e_spin, hipEventDisableTiming);
    hipEventCreateWithFlags(&e_block, hipEventDisableTiming | hipEventBlockingSync);
    const long long cyc = 500000;                            // wall_clock64 ticks at 100 MHz: 5 ms
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, (int*)nullptr); hipStreamSynchronize(s);
    const char* names[] = {"hipEventSynchronize(default event)", "hipEventSynchronize(blocking event)", "hipStreamSynchronize", "hipEventQuery + sleep 200 us"};
    printf("device flags: %s\n", flags_sel == 0 ? "default" : flags_sel == 1 ? "hipDeviceScheduleBlockingSync" : "hipDeviceScheduleYield");
    for (int mode = 0; mode < 4; ++mode) {
        double th = 0, pr = 0, wl = 0; const int reps = 20;
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, cyc, (int*)nullptr);
            hipEvent_t e = mode == 1 ? e_block : e_spin;
            hipEventRecord(e, s);
            const double t0 = now_thread(), p0 = now_proc(), w0 = now_wall();
            if (mode == 0 || mode == 1) hipEventSynchronize(e);
            else if (mode == 2) hipStreamSynchronize(s);
            else while (hipEventQuery(e) == hipErrorNotReady) std::this_thread::sleep_for(std::chrono::microseconds(200));
            th += now_thread() - t0; pr += now_proc() - p0; wl += now_wall() - w0;
        }
        printf("  %-40s wall %.2f ms  calling thread cpu %.2f ms  process cpu %.2f ms\n", names[mode], wl / reps * 1e3, th / reps * 1e3, pr / reps * 1e3);
    }
    return 0;
}
