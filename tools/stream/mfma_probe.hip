// mfma_probe.hip -- what the int8 matrix pipe of one SIMD sustains, by construction (round 5, config-3 kernel diagnosis).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
// 8 waves per workgroup (two per SIMD) or 4 (one per SIMD), one workgroup per CU (dynamic LDS keeps a second one out); every wave
// runs UNITS x 32 v_mfma_i32_32x32x32_i8.  Variants: accumulator chains per wave (1 = every MFMA depends on the previous one),
// an LDS fragment read (ds_read_b128) per MFMA, an s_barrier per unit.  Reports cycles per unit and wave (s_memtime) and the
// pipe's share: 32 cycles per MFMA at full rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));

__global__ void k_fill_random(unsigned *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + 12345u;
        x ^= x >> 16; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = x;
    }
}

template <int CHAINS, int READS, int BARRIER, int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_probe(int units, unsigned long long *out, int *sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63;
    i32x4_t q[32];
#pragma unroll
    for (int s = 0; s < 32; s++) q[s] = i32x4_t{(int)((lane * 2654435761u + s * 40503u) * 2246822519u), (int)((lane * 97u + s) * 3266489917u), (int)((lane + 13 * s) * 668265263u), (int)((lane * 31u + s * 7u) * 374761393u)};
#pragma unroll
    for (int s = 0; s < 32; s++) asm volatile("" : "+v"(q[s]));
    for (int i = threadIdx.x; i < 32 * 1040 / 4; i += blockDim.x) reinterpret_cast<int *>(lds)[i] = i * 2654435761u;
    __syncthreads();
    i32x16_t acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[c][i] = 0;
    const unsigned off = (unsigned)(lane & 31) * 1040u + (unsigned)(lane >> 5) * 16u;
    i32x4_t a[4];
#pragma unroll
    for (int f = 0; f < 4; f++) a[f] = READS ? *reinterpret_cast<const i32x4_t *>(lds + off + f * 32) : i32x4_t{lane, f, 3, 4};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int u = 0; u < units; u++) {
#pragma unroll
        for (int ks = 0; ks < 32; ks++) {
            acc[ks % CHAINS] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks % 4], q[ks], acc[ks % CHAINS], 0, 0, 0);
            if (READS) a[ks % 4] = *reinterpret_cast<const i32x4_t *>(lds + off + ((ks + 4) & 31) * 32);
        }
#pragma unroll
        for (int ks = 0; ks < 32; ks++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (READS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++)
#pragma unroll
        for (int i = 0; i < 16; i++) s += acc[c][i];
    if (s == 0x12345678) sink[0] = s;
    if (lane == 0) out[blockIdx.x * NW + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CHAINS, int READS, int BARRIER, int NW> static void run(const char *name, int units) {
    unsigned long long *d_out;
    int *d_sink;
    const int wgs = 256;
    hipMalloc(&d_out, wgs * NW * 8);
    hipMalloc(&d_sink, 64);
    auto kern = k_probe<CHAINS, READS, BARRIER, NW>;
    const int lds_bytes = 140 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(NW * 64), lds_bytes, 0, units, d_out, d_sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<unsigned long long> h(wgs * NW);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double)v;
    const double cyc = sum / h.size() / units;
    const double waves_per_simd = NW / 4.0;
    printf("%-44s %2d waves  %8.0f cycles per unit and wave  pipe share %.2f  wall %.3f ms  => %.0f TOP/s, clock %.2f GHz\n", name, NW, cyc,
           32.0 * 32.0 * waves_per_simd / cyc, best, 2.0 * 32 * 32 * 32 * 32.0 * units * NW * wgs / (best * 1e-3) / 1e12,
           cyc * units / (best * 1e-3) / 1e9);
    hipFree(d_out);
    hipFree(d_sink);
}


// ---- the same stream fed by an LDS-DMA ring (4 slots x 32 rows x 1040 B, units requested 3 ahead, rows must have landed one barrier
// early), rows streamed from a table in HBM.  DMODE: 1 = waves 0-3 request 8 rows each behind MFMAs 1..8, 2 = all 8 waves 4 rows each behind
// MFMAs 1..4, 3 = waves 0-3, one row every 4th MFMA, 4 = waves 0-3, all 8 rows in front of the first MFMA, 5 = wave 0-1 request 16 rows each
template <int DMODE, int MMA, int NW>
__global__ __launch_bounds__(NW * 64, 2) void k_probe_dma(const char *rows, unsigned n_rows, int units, unsigned long long *out, int *sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr int NS = 4, D = 3, UNIT = 32 * 1040;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    i32x4_t q[32];
#pragma unroll
    for (int s = 0; s < 32; s++) q[s] = i32x4_t{(int)((lane * 2654435761u + s * 40503u) * 2246822519u), (int)((lane * 97u + s) * 3266489917u), (int)((lane + 13 * s) * 668265263u), (int)((lane * 31u + s * 7u) * 374761393u)};
#pragma unroll
    for (int s = 0; s < 32; s++) asm volatile("" : "+v"(q[s]));
    i32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    const unsigned lane16 = (unsigned)lane * 16u;
    constexpr int NISS = DMODE == 2 ? 8 : (DMODE == 5 ? 2 : 4), IPW = 32 / NISS;
    const bool issuer = wave < NISS;
    unsigned ftile = blockIdx.x, fslot = 0;
    const unsigned step = gridDim.x;
    unsigned long long pbase = 0;
    unsigned plds = 0;
    auto begin = [&]() {
        const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((ftile % (n_rows / 32)) * 32));
        pbase = (unsigned long long)rows + (unsigned long long)(r0 + IPW * wave) * 1024ull;
        plds = lds_base + fslot * UNIT + (unsigned)(wave * IPW * 1040);
    };
    auto piece = [&]() {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(lane16), "s"(pbase), "s"(plds) : "memory");
        pbase += 1024;
        plds += 1040;
    };
    auto advance = [&]() {
        ftile += step;
        fslot = fslot + 1 == NS ? 0 : fslot + 1;
    };
    auto wait_units = [&](int u) {
        if (!issuer) return;
        if (u * IPW == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (u * IPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (u * IPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (u * IPW == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (issuer) {
        for (int u = 0; u < D; u++) {
            begin();
            for (int i = 0; i < IPW; i++) piece();
            advance();
        }
    }
    wait_units(D - 2);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned off = (unsigned)(lane & 31) * 1040u + (unsigned)(lane >> 5) * 16u;
    unsigned cslot = 0;
    i32x4_t a[4];
#pragma unroll
    for (int f = 0; f < 4; f++) a[f] = *reinterpret_cast<const i32x4_t *>(lds + off + f * 32);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int u = 0; u < units; u++) {
        const unsigned nslot = cslot + 1 == NS ? 0 : cslot + 1;
        if (issuer) begin();
        if (DMODE == 4 && issuer) {
#pragma unroll
            for (int i = 0; i < IPW; i++) piece();
        }
#pragma unroll
        for (int ks = 0; ks < 32; ks++) {
            if (MMA) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks % 4], q[ks], acc, 0, 0, 0);
            if (MMA) {
                const int f = ks + 4;
                a[ks % 4] = f < 32 ? *reinterpret_cast<const i32x4_t *>(lds + cslot * UNIT + off + f * 32)
                                   : *reinterpret_cast<const i32x4_t *>(lds + nslot * UNIT + off + (f - 32) * 32);
            }
            if (issuer) {
                if ((DMODE == 1 || DMODE == 2 || DMODE == 5) && ks >= 1 && ks <= IPW) piece();
                if (DMODE == 3 && (ks & 3) == 1) piece();
            }
        }
        cslot = nslot;
        if (issuer) advance();
        wait_units(D - 2);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i];
    if (s == 0x12345678) sink[0] = s;
    if (lane == 0) out[blockIdx.x * NW + wave] = t1 - t0;
}

template <int DMODE, int MMA> static void run_dma(const char *name, const char *rows, unsigned n_rows) {
    constexpr int NW = 8;
    unsigned long long *d_out;
    int *d_sink;
    const int wgs = 256;
    const int units = (int)(n_rows / 32 / wgs);
    hipMalloc(&d_out, wgs * NW * 8);
    hipMalloc(&d_sink, 64);
    auto kern = k_probe_dma<DMODE, MMA, NW>;
    const int lds_bytes = 4 * 32 * 1040 + 4096;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(NW * 64), lds_bytes, 0, rows, n_rows, units, d_out, d_sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)units * wgs * 32 * 1024;
    printf("%-58s wall %.3f ms  => %.0f GB/s, %.0f TOP/s (as 256 queries)\n", name, best, bytes / (best * 1e-3) / 1e9,
           2.0 * bytes * 256 / (best * 1e-3) / 1e12);
    hipFree(d_out);
    hipFree(d_sink);
}

// ---- 4 waves x 64 queries: one wave per SIMD, every A fragment feeds TWO MFMAs (half the LDS reads per MAC); all four waves request rows
// (8 each, one behind every PSTEP-th k-step).  SCREEN: 16 VALU per unit (running maxima of the previous unit's accumulators) in the stream.
template <int MMA, int PSTEP, int SCREEN>
__global__ __launch_bounds__(256, 1) void k_probe_dma64(const char *rows, unsigned n_rows, int units, unsigned long long *out, int *sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr int NS = 4, D = 3, UNIT = 32 * 1040, IPW = 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    i32x4_t q[64];
#pragma unroll
    for (int s = 0; s < 64; s++) q[s] = i32x4_t{(int)((lane * 2654435761u + s * 40503u) * 2246822519u), (int)((lane * 97u + s) * 3266489917u), (int)((lane + 13 * s) * 668265263u), (int)((lane * 31u + s * 7u) * 374761393u)};
#pragma unroll
    for (int s = 0; s < 64; s++) asm volatile("" : "+v"(q[s]));
    i32x16_t acc0, acc1, p0, p1;
#pragma unroll
    for (int i = 0; i < 16; i++) acc0[i] = 0, acc1[i] = 0, p0[i] = 0, p1[i] = 0;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned ftile = blockIdx.x, fslot = 0;
    const unsigned step = gridDim.x;
    unsigned long long pbase = 0;
    unsigned plds = 0;
    auto begin = [&]() {
        const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((ftile % (n_rows / 32)) * 32));
        pbase = (unsigned long long)rows + (unsigned long long)(r0 + IPW * wave) * 1024ull;
        plds = lds_base + fslot * UNIT + (unsigned)(wave * IPW * 1040);
    };
    auto piece = [&]() {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(lane16), "s"(pbase), "s"(plds) : "memory");
        pbase += 1024;
        plds += 1040;
    };
    auto advance = [&]() {
        ftile += step;
        fslot = fslot + 1 == NS ? 0 : fslot + 1;
    };
    for (int u = 0; u < D; u++) {
        begin();
        for (int i = 0; i < IPW; i++) piece();
        advance();
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned off = (unsigned)(lane & 31) * 1040u + (unsigned)(lane >> 5) * 16u;
    unsigned cslot = 0;
    i32x4_t a[4];
#pragma unroll
    for (int f = 0; f < 4; f++) a[f] = *reinterpret_cast<const i32x4_t *>(lds + off + f * 32);
    int m = 0;
    for (int u = 0; u < units; u++) {
        const unsigned nslot = cslot + 1 == NS ? 0 : cslot + 1;
        begin();
#pragma unroll
        for (int ks = 0; ks < 32; ks++) {
            if (MMA) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks % 4], q[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks % 4], q[32 + ks], acc1, 0, 0, 0);
                const int f = ks + 4;
                a[ks % 4] = f < 32 ? *reinterpret_cast<const i32x4_t *>(lds + cslot * UNIT + off + f * 32)
                                   : *reinterpret_cast<const i32x4_t *>(lds + nslot * UNIT + off + (f - 32) * 32);
            }
            if (ks >= 1 && (ks - 1) % PSTEP == 0 && (ks - 1) / PSTEP < IPW) piece();
            if (SCREEN && ks >= 8 && ks < 24) {
                const int r = ks - 8;
                m = max(max(m, p0[r]), p1[r]);
            }
        }
        if (SCREEN) {
            if (__ballot(m >= 0x7ffffff0) != 0) sink[1] = m;
            p0 = acc0;
            p1 = acc1;
#pragma unroll
            for (int i = 0; i < 16; i++) acc0[i] = 0, acc1[i] = 0;
        }
        cslot = nslot;
        advance();
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int s = m;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i];
    if (s == 0x12345678) sink[0] = s;
}

template <int MMA, int PSTEP, int SCREEN> static void run_dma64(const char *name, const char *rows, unsigned n_rows) {
    unsigned long long *d_out;
    int *d_sink;
    const int wgs = 256;
    const int units = (int)(n_rows / 32 / wgs);
    hipMalloc(&d_out, wgs * 8 * 8);
    hipMalloc(&d_sink, 64);
    auto kern = k_probe_dma64<MMA, PSTEP, SCREEN>;
    const int lds_bytes = 4 * 32 * 1040 + 4096;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds_bytes, 0, rows, n_rows, units, d_out, d_sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)units * wgs * 32 * 1024;
    printf("%-58s wall %.3f ms  => %.0f GB/s, %.0f TOP/s (as 256 queries)\n", name, best, bytes / (best * 1e-3) / 1e9,
           2.0 * bytes * 256 / (best * 1e-3) / 1e12);
    hipFree(d_out);
    hipFree(d_sink);
}

// ---- how fast can CUs ingest rows that another CU of the same XCD streams too (two query tiles of a wide-row batch)?  Ring DMA only,
// `share` workgroups walk the SAME tile sequence (ids 8 apart land on one XCD); the table crosses HBM once, L2 / MALL serve the rest.
template <int NW> __global__ __launch_bounds__(NW * 64, 2) void k_probe_share(const char *rows, unsigned n_rows, int units, int share, int *sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr int NS = 4, D = 3, UNIT = 32 * 1040, IPW = 32 / NW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    const unsigned lane16 = (unsigned)lane * 16u;
    // workgroup b: stream id = (b / (8 * share)) * 8 + b % 8 -- the `share` workgroups b, b + 8, ... of one XCD slot walk one stream
    const unsigned b = blockIdx.x, stream = (b / (8u * share)) * 8u + b % 8u, n_streams = gridDim.x / share;
    unsigned ftile = stream, fslot = 0;
    unsigned long long pbase = 0;
    unsigned plds = 0;
    auto begin = [&]() {
        const unsigned r0 = (unsigned)__builtin_amdgcn_readfirstlane((int)((ftile % (n_rows / 32)) * 32));
        pbase = (unsigned long long)rows + (unsigned long long)(r0 + IPW * wave) * 1024ull;
        plds = lds_base + fslot * UNIT + (unsigned)(wave * IPW * 1040);
    };
    auto piece = [&]() {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(pbase), "s"(plds) : "memory");
        pbase += 1024;
        plds += 1040;
    };
    for (int u = 0; u < D; u++) {
        begin();
        for (int i = 0; i < IPW; i++) piece();
        ftile += n_streams;
        fslot = fslot + 1 == NS ? 0 : fslot + 1;
    }
    for (int u = 0; u < units; u++) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (32 / NW)) : "memory");
        asm volatile("s_barrier" ::: "memory");
        begin();
#pragma unroll
        for (int i = 0; i < IPW; i++) piece();
        ftile += n_streams;
        fslot = fslot + 1 == NS ? 0 : fslot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (reinterpret_cast<int *>(lds)[lane] == 0x12345678) sink[0] = 1;
}
static void run_share(const char *name, const char *rows, unsigned n_rows, int share, int wgs) {
    int *d_sink;
    hipMalloc(&d_sink, 64);
    auto kern = k_probe_share<4>;
    const int lds_bytes = 4 * 32 * 1040;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    const int units = (int)((size_t)n_rows / 32 / (wgs / share));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds_bytes, 0, rows, n_rows, units, share, d_sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double table = (double)units * (wgs / share) * 32 * 1024;
    printf("%-64s wall %.3f ms  => table %.0f GB/s, LDS-DMA ingest %.0f GB/s (%.1f B/clk/CU at 2.1 GHz)\n", name, best, table / (best * 1e-3) / 1e9,
           table * share / (best * 1e-3) / 1e9, table * share / (best * 1e-3) / 256 / 2.1e9);
    hipFree(d_sink);
}

int main() {
    const int U = 2000;
    if (!getenv("PROBE_SHARE_ONLY")) {
    run<1, 0, 0, 8>("1 chain, no reads, no barrier", U);
    run<2, 0, 0, 8>("2 chains", U);
    run<4, 0, 0, 8>("4 chains", U);
    run<1, 0, 0, 4>("1 chain, one wave per SIMD", U);
    run<2, 0, 0, 4>("2 chains, one wave per SIMD", U);
    run<4, 0, 0, 4>("4 chains, one wave per SIMD", U);
    run<1, 1, 0, 8>("1 chain + ds_read_b128 per MFMA", U);
    run<2, 1, 0, 8>("2 chains + ds_read_b128 per MFMA", U);
    run<1, 1, 1, 8>("1 chain + reads + barrier per unit", U);
    run<2, 1, 1, 8>("2 chains + reads + barrier per unit", U);
    run<1, 1, 0, 4>("1 chain + reads, one wave per SIMD", U);
    run<2, 1, 0, 4>("2 chains + reads, one wave per SIMD", U);
    }
    {
        const unsigned n_rows = 8u << 20;   // 8 Mi rows x 1 KiB
        char *rows;
        hipMalloc(&rows, (size_t)n_rows * 1024 + 4096);
        hipLaunchKernelGGL(k_fill_random, dim3(65536), dim3(256), 0, 0, (unsigned *)rows, (size_t)n_rows * 256);
        hipDeviceSynchronize();
        run_dma<1, 0>("ring DMA only, waves 0-3 x 8 rows", rows, n_rows);
        run_dma<2, 0>("ring DMA only, 8 waves x 4 rows", rows, n_rows);
        run_dma<1, 1>("MFMA + reads + DMA: waves 0-3 x 8 rows behind MFMAs 1..8", rows, n_rows);
        run_dma<2, 1>("MFMA + reads + DMA: 8 waves x 4 rows behind MFMAs 1..4", rows, n_rows);
        run_dma<3, 1>("MFMA + reads + DMA: waves 0-3, a row every 4th MFMA", rows, n_rows);
        run_dma<4, 1>("MFMA + reads + DMA: waves 0-3, 8 rows before MFMA 0", rows, n_rows);
        run_dma<5, 1>("MFMA + reads + DMA: waves 0-1 x 16 rows behind MFMAs 1..16", rows, n_rows);
        run_dma64<1, 1, 0>("4 waves x 64 queries: rows behind MFMA pairs 1..8", rows, n_rows);
        run_dma64<1, 2, 0>("4 waves x 64 queries: a row every 2nd pair", rows, n_rows);
        run_dma64<1, 4, 0>("4 waves x 64 queries: a row every 4th pair", rows, n_rows);
        run_dma64<1, 2, 1>("4 waves x 64 queries: every 2nd pair + screening", rows, n_rows);
        run_share("DMA only, 256 workgroups, every row once", rows, n_rows, 1, 256);
        run_share("DMA only, 512 workgroups (2 per CU), every row once", rows, n_rows, 1, 512);
        run_share("DMA only, 256 workgroups, pairs share a row stream (2 x)", rows, n_rows, 2, 256);
        run_share("DMA only, 512 workgroups, pairs share a row stream (2 x)", rows, n_rows, 2, 512);
        run_share("DMA only, 512 workgroups, four share a row stream (4 x)", rows, n_rows, 4, 512);
        hipFree(rows);
    }
    return 0;
}
