// overlap_probe.hip -- what may run BESIDE a table scan on MI355X, measured before touching the product kernels (round 4).
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream/overlap_probe tools/stream/overlap_probe.hip && tools/stream/overlap_probe
//
// Questions:
//  1. hipExtStreamCreateWithCUMask: where do workgroups of a masked stream land (XCC, SE, CU from the hardware id registers)?
//  2. How fast does a streaming read run on 256 / 248 / 240 / 224 CUs (a scan that leaves CUs to the small kernels)?
//  3. A second stream reading other memory on the reserved CUs at the same time: what does each side get?
//  4. Tile dealing: static grid-stride against a shared counter (vector atomic by one lane, and s_atomic_add), alone and with
//     an interfering kernel that occupies every CU for part of the run (late / slowed workgroups -> a tail for the static deal).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr size_t TILE_BYTES = 49152;   // 16 rows x 3 KiB, the fp32 filter's tile
constexpr int TILE_U4 = TILE_BYTES / 16;

__device__ __forceinline__ uint32_t hw_id() { return __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); }
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)); }

__global__ void k_where(uint32_t *out) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw_id();
        out[2 * blockIdx.x + 1] = xcc_id();
    }
    // stay a little so that the workgroups spread over the CUs the mask allows
    uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 200000) {}
}

// MODE 0: static grid-stride over tiles; 1: shared counter, vector atomic by lane 0 of wave 0 + LDS broadcast, one tile ahead;
// 2: the same through s_atomic_add (scalar cache path, lgkmcnt: no VM counter involved)
template <int MODE>
__global__ __launch_bounds__(256) void k_tiles(const uint4 *p, uint32_t n_tiles, uint32_t *counter, uint32_t *sink, uint32_t *per_wg) {
    __shared__ uint32_t next_s[2];
    uint32_t acc = 0, done = 0;
    uint32_t tile, nxt = 0;
    if (MODE == 0) tile = blockIdx.x;
    else {
        if (threadIdx.x == 0) {
            if (MODE == 1) next_s[0] = atomicAdd(counter, 1u);
            else {
                uint32_t v;
                asm volatile("s_atomic_add %0, %1, %2 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(counter), "s"(1u) : "memory");
                next_s[0] = v;
            }
        }
        __syncthreads();
        tile = next_s[0];
    }
    uint32_t par = 0;
    while (tile < n_tiles) {
        if (MODE != 0 && threadIdx.x == 0) {   // fetch the tile after this one while this one streams
            if (MODE == 1) nxt = atomicAdd(counter, 1u);
            else asm volatile("s_atomic_add %0, %1, %2 glc" : "=s"(nxt) : "s"(counter), "s"(1u) : "memory");
        }
        const uint4 *q = p + (size_t)tile * TILE_U4 + threadIdx.x;
        uint4 v[12];
#pragma unroll
        for (int u = 0; u < 12; u++) {
            v[u].x = __builtin_nontemporal_load(&q[u * 256].x); v[u].y = __builtin_nontemporal_load(&q[u * 256].y);
            v[u].z = __builtin_nontemporal_load(&q[u * 256].z); v[u].w = __builtin_nontemporal_load(&q[u * 256].w);
        }
#pragma unroll
        for (int u = 0; u < 12; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        done++;
        if (MODE == 0) tile += gridDim.x;
        else {
            if (threadIdx.x == 0) {
                if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                next_s[par ^ 1] = nxt;
            }
            __syncthreads();
            par ^= 1;
            tile = next_s[par];
        }
    }
    if (acc == 0x12345678u) *sink = acc;
    if (threadIdx.x == 0 && per_wg) per_wg[blockIdx.x] = done;
}

// occupies the machine with register-heavy busy workgroups for ~`cycles` each (an "aux kernel" that holds CUs when a scan starts)
__global__ __launch_bounds__(256) void k_busy(uint64_t cycles, uint32_t *sink) {
    uint64_t t0 = wall_clock64();   // 100 MHz
    uint32_t a = threadIdx.x;
    while (wall_clock64() - t0 < cycles) a = a * 1664525u + 1013904223u;
    if (a == 0x12345678u) *sink = a;
}

static hipStream_t masked_stream(const std::vector<uint32_t> &mask) {
    hipStream_t s;
    CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    return s;
}
static std::vector<uint32_t> mask_range(int lo, int hi, int total) {   // bits [lo, hi)
    std::vector<uint32_t> m((total + 31) / 32, 0u);
    for (int i = lo; i < hi; i++) m[i / 32] |= 1u << (i % 32);
    return m;
}
static std::vector<uint32_t> mask_not(const std::vector<uint32_t> &a, int total) {
    std::vector<uint32_t> m(a.size(), 0u);
    for (int i = 0; i < total; i++)
        if (!(a[i / 32] >> (i % 32) & 1u)) m[i / 32] |= 1u << (i % 32);
    return m;
}



int main(int argc, char **argv) {
    hipDeviceProp_t pr;
    CHECK(hipGetDeviceProperties(&pr, 0));
    const int cu = pr.multiProcessorCount;
    printf("# %s, %d CUs\n", pr.name, cu);
    const size_t bytes = (size_t)12 << 30;
    const uint32_t n_tiles = (uint32_t)(bytes / TILE_BYTES);
    uint4 *d, *d2;
    uint32_t *sink, *counter, *where, *per_wg;
    CHECK(hipMalloc(&d, bytes));
    CHECK(hipMalloc(&d2, (size_t)2 << 30));
    CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&counter, 256)); CHECK(hipMalloc(&where, 8 * 4096)); CHECK(hipMalloc(&per_wg, 4 * 8192));
    CHECK(hipMemset(d, 1, bytes)); CHECK(hipMemset(d2, 1, (size_t)2 << 30));
    hipEvent_t e0, e1, f0, f1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&f0)); CHECK(hipEventCreate(&f1));

    // ---- 1. where do masked workgroups land
    auto report_where = [&](const char *tag, hipStream_t s, int wgs) {
        hipLaunchKernelGGL(k_where, dim3(wgs), dim3(64), 0, s, where);
        CHECK(hipStreamSynchronize(s));
        std::vector<uint32_t> h(2 * wgs);
        CHECK(hipMemcpy(h.data(), where, 8 * wgs, hipMemcpyDeviceToHost));
        int per_xcc[16] = {0};
        std::vector<int> seen(16 * 4096, 0);
        int distinct = 0;
        for (int i = 0; i < wgs; i++) {
            const uint32_t id = h[2 * i], x = h[2 * i + 1] & 15;
            const uint32_t cuid = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7;
            per_xcc[x]++;
            const int key = (int)(x * 4096 + se * 64 + sh * 16 + cuid);
            if (!seen[key]++) distinct++;
        }
        printf("%-28s %4d wgs on %3d distinct (xcc,se,sh,cu); per xcc:", tag, wgs, distinct);
        for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
        printf("\n");
    };
    hipStream_t s_all;
    CHECK(hipStreamCreate(&s_all));
    report_where("unmasked", s_all, 1024);
    for (int R : {8, 16, 32}) {
        auto rsv = mask_range(cu - R, cu, cu);
        hipStream_t sr = masked_stream(rsv), sm = masked_stream(mask_not(rsv, cu));
        char tag[64];
        snprintf(tag, sizeof tag, "mask last %d bits", R);
        report_where(tag, sr, 256);
        snprintf(tag, sizeof tag, "mask all but last %d", R);
        report_where(tag, sm, 1024);
        // every 256/R-th bit instead
        std::vector<uint32_t> spread((cu + 31) / 32, 0u);
        for (int i = 0; i < R; i++) { int b = i * (cu / R); spread[b / 32] |= 1u << (b % 32); }
        hipStream_t ss = masked_stream(spread);
        snprintf(tag, sizeof tag, "mask %d spread bits", R);
        report_where(tag, ss, 256);
        CHECK(hipStreamDestroy(sr)); CHECK(hipStreamDestroy(sm)); CHECK(hipStreamDestroy(ss));
    }

    // ---- 2 + 3. streaming read on fewer CUs, alone and beside a reader on the reserved CUs
    auto run_read = [&](hipStream_t s, int grid, const uint4 *buf, uint32_t tiles, hipEvent_t a, hipEvent_t b) {
        CHECK(hipEventRecord(a, s));
        hipLaunchKernelGGL((k_tiles<0>), dim3(grid), dim3(256), 0, s, buf, tiles, counter, sink, (uint32_t *)nullptr);
        CHECK(hipEventRecord(b, s));
    };
    auto ms_of = [&](hipEvent_t a, hipEvent_t b) { float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; };
    for (int spread_mode = 0; spread_mode < 2; spread_mode++)
    for (int R : {0, 8, 16, 32}) {
        std::vector<uint32_t> rsv;
        if (spread_mode == 0) rsv = mask_range(cu - R, cu, cu);
        else { rsv.assign((cu + 31) / 32, 0u); for (int i = 0; i < R; i++) { int b = i * (cu / std::max(R, 1)); rsv[b / 32] |= 1u << (b % 32); } }
        if (R == 0 && spread_mode == 1) continue;
        hipStream_t sm = R ? masked_stream(mask_not(rsv, cu)) : s_all;
        hipStream_t sr = R ? masked_stream(rsv) : nullptr;
        const int grid = (cu - R) * 2;
        float best = 1e30f;
        for (int r = 0; r < 4; r++) { run_read(sm, grid, d, n_tiles, e0, e1); CHECK(hipStreamSynchronize(sm)); best = std::min(best, ms_of(e0, e1)); }
        printf("read 12 GiB on %3d CUs (%s): %.3f ms = %.0f GB/s", cu - R, spread_mode ? "spread" : "last", best, bytes / (best * 1e-3) / 1e9);
        if (R) {
            // beside it: 640 MB read on the reserved CUs, started 0.1 ms into the scan
            const uint32_t aux_tiles = (uint32_t)((size_t)640000000 / TILE_BYTES);
            float bm = 1e30f, ba = 1e30f;
            for (int r = 0; r < 4; r++) {
                run_read(sm, grid, d, n_tiles, e0, e1);
                run_read(sr, R * 2, d2, aux_tiles, f0, f1);
                CHECK(hipStreamSynchronize(sm)); CHECK(hipStreamSynchronize(sr));
                bm = std::min(bm, ms_of(e0, e1)); ba = std::min(ba, ms_of(f0, f1));
            }
            printf("   | with 640 MB on the %d reserved: main %.3f ms (%.0f GB/s), aux %.3f ms (%.0f GB/s)", R, bm, bytes / (bm * 1e-3) / 1e9, ba, 640e6 / (ba * 1e-3) / 1e9);
            // the aux alone
            float bo = 1e30f;
            for (int r = 0; r < 3; r++) { run_read(sr, R * 2, d2, aux_tiles, f0, f1); CHECK(hipStreamSynchronize(sr)); bo = std::min(bo, ms_of(f0, f1)); }
            printf(", aux alone %.3f ms", bo);
            CHECK(hipStreamDestroy(sm)); CHECK(hipStreamDestroy(sr));
        }
        printf("\n");
    }
    // unmasked: main scan with an unmasked 640 MB aux read in a second stream (what 'just overlap them' costs)
    {
        hipStream_t s2; CHECK(hipStreamCreate(&s2));
        const uint32_t aux_tiles = (uint32_t)((size_t)640000000 / TILE_BYTES);
        float bm = 1e30f, ba = 1e30f;
        for (int r = 0; r < 4; r++) {
            run_read(s_all, cu * 2, d, n_tiles, e0, e1);
            run_read(s2, 64, d2, aux_tiles, f0, f1);
            CHECK(hipDeviceSynchronize());
            bm = std::min(bm, ms_of(e0, e1)); ba = std::min(ba, ms_of(f0, f1));
        }
        printf("unmasked main + unmasked 64-wg aux read: main %.3f ms (%.0f GB/s), aux %.3f ms\n", bm, bytes / (bm * 1e-3) / 1e9, ba);
    }

    // ---- 4. tile dealing
    auto run_mode = [&](int mode, int grid, bool interfere, const char *tag) {
        hipStream_t s2; CHECK(hipStreamCreate(&s2));
        float best = 1e30f, worst = 0;
        uint32_t mn = 0, mx = 0;
        for (int r = 0; r < 4; r++) {
            CHECK(hipMemsetAsync(counter, 0, 4, s_all));
            CHECK(hipStreamSynchronize(s_all));
            if (interfere) hipLaunchKernelGGL(k_busy, dim3(cu * 2), dim3(256), 0, s2, (uint64_t)30000 /* x 100 MHz counter ~ 0.3 ms */, sink);
            CHECK(hipEventRecord(e0, s_all));
            if (mode == 0) hipLaunchKernelGGL((k_tiles<0>), dim3(grid), dim3(256), 0, s_all, d, n_tiles, counter, sink, per_wg);
            else if (mode == 1) hipLaunchKernelGGL((k_tiles<1>), dim3(grid), dim3(256), 0, s_all, d, n_tiles, counter, sink, per_wg);
            else hipLaunchKernelGGL((k_tiles<2>), dim3(grid), dim3(256), 0, s_all, d, n_tiles, counter, sink, per_wg);
            CHECK(hipEventRecord(e1, s_all));
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("%s: kernel failed: %s\n", tag, hipGetErrorString(e)); return; }
            const float ms = ms_of(e0, e1);
            best = std::min(best, ms); worst = std::max(worst, ms);
            std::vector<uint32_t> h(grid);
            CHECK(hipMemcpy(h.data(), per_wg, 4 * grid, hipMemcpyDeviceToHost));
            uint64_t tot = 0;
            mn = ~0u; mx = 0;
            for (auto v : h) { tot += v; mn = std::min(mn, v); mx = std::max(mx, v); }
            if (tot != n_tiles) { printf("%s: tiles processed %llu != %u (dealing broken)\n", tag, (unsigned long long)tot, n_tiles); return; }
        }
        printf("%-44s best %.3f ms = %.0f GB/s, worst %.3f ms; tiles per wg %u..%u\n", tag, best, bytes / (best * 1e-3) / 1e9, worst, mn, mx);
        CHECK(hipStreamDestroy(s2));
    };
    run_mode(0, cu * 2, false, "static deal, alone");
    run_mode(1, cu * 2, false, "counter (vector atomic), alone");
    if (argc > 1 && !strcmp(argv[1], "satomic")) run_mode(2, cu * 2, false, "counter (s_atomic_add), alone");
    run_mode(0, cu * 2, true, "static deal, busy kernel holding the CUs");
    run_mode(1, cu * 2, true, "counter (vector atomic), busy kernel");
    if (argc > 1 && !strcmp(argv[1], "satomic")) run_mode(2, cu * 2, true, "counter (s_atomic_add), busy kernel");
    return 0;
}
