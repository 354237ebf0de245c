// stream_read.hip -- device read-bandwidth ceiling of the box, for context next to roofline.frac (SURVEY.md 8d asks for a
// measured stream-read peak beside the 8 TB/s vendor figure).  Reads a 30.72 GB buffer (the headline workload's size) once
// per launch with 16-byte loads, UNROLL of them in flight per lane, plain and non-temporal, at a few grid sizes.
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream/stream_read tools/stream/stream_read.hip && tools/stream/stream_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void k_read(const uint4 *p, size_t n16, uint32_t *sink) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint4 *q = p + i + u * stride;
            if (NT) {
                v[u].x = __builtin_nontemporal_load(&q->x); v[u].y = __builtin_nontemporal_load(&q->y);
                v[u].z = __builtin_nontemporal_load(&q->z); v[u].w = __builtin_nontemporal_load(&q->w);
            } else v[u] = *q;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;  // keeps the loads alive
}

template <bool NT, int UNROLL> static double run(const uint4 *d, size_t n16, uint32_t *sink, int grid, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_read<NT, UNROLL>), dim3(grid), dim3(256), 0, 0, d, n16, sink);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_read<NT, UNROLL>), dim3(grid), dim3(256), 0, 0, d, n16, sink);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return (double)n16 * 16 / (best * 1e-3) / 1e9;
}

int main() {
    const size_t bytes = 30720000000ull;
    const size_t n16 = bytes / 16;
    uint4 *d; uint32_t *sink;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, bytes);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cu = pr.multiProcessorCount;
    printf("# %s, %d CUs, buffer %.2f GB, best of 5 launches, GB/s\n", pr.name, cu, bytes / 1e9);
    for (int wg : {2, 4, 8, 16}) {
        const int grid = cu * wg;
        printf("wg/cu %2d: plain x4 %.0f  plain x8 %.0f  nt x4 %.0f  nt x8 %.0f\n", wg, run<false, 4>(d, n16, sink, grid, 5),
               run<false, 8>(d, n16, sink, grid, 5), run<true, 4>(d, n16, sink, grid, 5), run<true, 8>(d, n16, sink, grid, 5));
    }
    return 0;
}
