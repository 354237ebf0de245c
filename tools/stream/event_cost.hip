// event_cost.hip -- what a timing event between two kernels of one stream costs on the GPU's timeline, and whether events
// attached to the launch itself (hipExtLaunchKernelGGL start / stop events) are cheaper.
//   hipcc --offload-arch=gfx950 -O2 -o event_cost event_cost.hip && ./event_cost
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void k_spin(unsigned *p, int n) {
    unsigned v = threadIdx.x;
    for (int i = 0; i < n; i++) v = v * 1664525u + 1013904223u;
    if (v == 0x12345u) *p = v;
}
#define CHK(x)                                                                  \
    do {                                                                        \
        hipError_t e_ = (x);                                                    \
        if (e_ != hipSuccess) {                                                 \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                      \
            return 1;                                                           \
        }                                                                       \
    } while (0)

int main() {
    unsigned *d;
    CHK(hipMalloc(&d, 4));
    hipStream_t s, s2;
    CHK(hipStreamCreate(&s));
    CHK(hipStreamCreate(&s2));
    hipEvent_t ea, eb, ec, ed, nt;
    CHK(hipEventCreate(&ea));
    CHK(hipEventCreate(&eb));
    CHK(hipEventCreate(&ec));
    CHK(hipEventCreate(&ed));
    CHK(hipEventCreateWithFlags(&nt, hipEventDisableTiming));
    // round 6: the same pair with the release scope spelled out -- a default event releases to the SYSTEM (a cache write-back the
    // host needs to see device writes); these do not
    hipEvent_t fa, fb, ga, gb;
    CHK(hipEventCreateWithFlags(&fa, hipEventDisableSystemFence));
    CHK(hipEventCreateWithFlags(&fb, hipEventDisableSystemFence));
    CHK(hipEventCreateWithFlags(&ga, hipEventReleaseToDevice));
    CHK(hipEventCreateWithFlags(&gb, hipEventReleaseToDevice));
    const int CHAIN = 6, ITERS = 2000, SPIN = 200;
    auto run = [&](int mode) -> double {
        for (int w = 0; w < 2; w++) {
            auto t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < ITERS; it++) {
                for (int j = 0; j < CHAIN; j++) {
                    const bool mid = j == CHAIN / 2;
                    if (mode == 1 && mid) (void)hipEventRecord(ea, s);
                    if (mode == 6 && mid) (void)hipEventRecord(fa, s);
                    if (mode == 7 && mid) (void)hipEventRecord(ga, s);
                    if (mode == 3 && mid) (void)hipEventRecord(nt, s);
                    if (mode == 2 && mid) hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, ea, eb, 0, d, SPIN);
                    else if (mode == 4 && mid) hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, nullptr, eb, 0, d, SPIN);
                    else hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, d, SPIN);
                    if (mode == 1 && mid) (void)hipEventRecord(eb, s);
                    if (mode == 6 && mid) (void)hipEventRecord(fb, s);
                    if (mode == 7 && mid) (void)hipEventRecord(gb, s);
                    if (mode == 5 && mid) {   // another stream waits for the stop event of this launch
                        (void)hipEventRecord(ec, s);
                        (void)hipStreamWaitEvent(s2, ec, 0);
                        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, d, 1);
                    }
                }
                (void)hipStreamSynchronize(s);
            }
            (void)hipDeviceSynchronize();
            auto t1 = std::chrono::steady_clock::now();
            if (w == 1) return std::chrono::duration<double, std::micro>(t1 - t0).count() / ITERS;
        }
        return 0;
    };
    const char *names[] = {"plain chain of 6 kernels + sync", "two hipEventRecord around kernel 3", "hipExtLaunchKernelGGL(start, stop) on kernel 3",
                           "one no-timing hipEventRecord before kernel 3", "hipExtLaunchKernelGGL(null, stop) on kernel 3",
                           "record + other stream waits + its kernel", "two hipEventRecord, hipEventDisableSystemFence",
                           "two hipEventRecord, hipEventReleaseToDevice"};
    for (int m = 0; m < 8; m++) {
        const double us = run(m);
        float ms = -1;
        if (m == 1 || m == 2) (void)hipEventElapsedTime(&ms, ea, eb);
        if (m == 6) (void)hipEventElapsedTime(&ms, fa, fb);
        if (m == 7) (void)hipEventElapsedTime(&ms, ga, gb);
        printf("%-58s %8.2f us per iteration   (kernel 3 by its events: %.2f us)\n", names[m], us, ms * 1000.f);
    }
    // the stop event of an ext launch as a cross-stream dependency: does the waiter see the kernel's end?
    {
        unsigned *h;
        CHK(hipHostMalloc(&h, 4, hipHostMallocMapped));
        *h = 0;
        hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, nullptr, eb, 0, d, 2000000);
        CHK(hipStreamWaitEvent(s2, eb, 0));
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, d, 1);
        auto t0 = std::chrono::steady_clock::now();
        CHK(hipStreamSynchronize(s2));
        auto t1 = std::chrono::steady_clock::now();
        CHK(hipStreamSynchronize(s));
        auto t2 = std::chrono::steady_clock::now();
        printf("waiter stream finished after %.1f us, the long kernel's stream %.1f us later (waiter must not finish first: ~0 expected)\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count());
    }
    return 0;
}
