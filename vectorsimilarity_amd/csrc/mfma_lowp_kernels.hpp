// mfma_lowp_kernels.hpp -- MFMA filter for bf16 / fp16 / int8 rows (BASELINE configs 3 and 4 shapes).
//
// Same skeleton as k_mfma_filter (mfma_kernels.hpp): rows stream HBM -> LDS once by non-temporal
// LDS-DMA into a 3-slot ring of 16 KiB (counted vmcnt + one raw barrier per slot, XOR-swizzled image
// so the A-operand ds_read_b128 are conflict free), queries stay in VGPRs as MFMA B operands, the
// epilogue turns dots into scores and either records per-tile minima (probe) or emits candidates.
// Differences from the fp32 kernel:
//   * the stored elements ARE the MFMA operands (no conversion): 16 B per lane per k-step,
//     v_mfma_f32_16x16x32_{bf16,f16} (32 elements) or v_mfma_i32_16x16x64_i8 (64 elements);
//   * 8 waves per workgroup, 16*NQW queries per wave: 128 queries per pass for bf16/fp16, 256 for int8,
//     so config-4 (batch 128) and config-3 (batch 256) batches read every row exactly once;
//   * bf16/fp16: products are exact in fp32, only the MFMA's accumulation order differs from the
//     reference's lane order, so the bound E is ~2^-22*d*|x||q| -- a few hundred survivors per query --
//     and the survivors are re-scored by k_exact_pairs in the reference order;
//   * int8: the int32 dot is exact and order free, so the kernel computes the reference score itself
//     (L2.cpp:164-174, IP.cpp:258-271): E = 0 and there is no re-rank.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfma_kernels.hpp"

namespace vsg {

enum LowpKind { LP_BF16 = 0, LP_F16 = 1, LP_I8 = 2 };
enum LowpEpi { LE_FP_L2 = 0, LE_FP_IP = 1, LE_I8_L2 = 2, LE_I8_IP = 3, LE_I8_COS = 4 };

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

template <int LK> struct LowpOps;
template <> struct LowpOps<LP_BF16> {
    using acc_t = f32x4_t;
    __device__ static inline acc_t mma(u32x4_t a, u32x4_t b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct LowpOps<LP_F16> {
    using acc_t = f32x4_t;
    __device__ static inline acc_t mma(u32x4_t a, u32x4_t b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct LowpOps<LP_I8> {
    using acc_t = i32x4_t;
    __device__ static inline acc_t mma(u32x4_t a, u32x4_t b, acc_t c) {
        return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_t, a), __builtin_bit_cast(i32x4_t, b), c, 0, 0, 0);
    }
};

struct LowpParams {
    const char *const *slabs;            // row slabs
    const uint32_t *const *aux_slabs;    // per-row 4-byte aux: float |x|^2 (fp), int32 sum x^2 (int8 L2), float norm (int8 Cosine)
    uint32_t slab_shift, slab_mask;
    uint32_t row_stride;                 // bytes between rows (dim*elem, +4 for int8 Cosine)
    uint32_t n_rows;
    uint32_t tile_first, tile_step, n_tiles;   // tile t covers rows (tile_first + t*tile_step)*RT ...
    const uint4 *qfrag;                  // [q_tile][wave][NQW][KSTEPS][lane] 16-B B-operand fragments
    const uint32_t *qaux;                // per query: float |q|^2 | int32 sum q^2 | float norm
    int epi;
    float cE, absE;
    float *tilemin;                      // MF_PROBE: [queries][tilemin_stride]
    size_t tilemin_stride;
    const float *tau;                    // MF_FILTER
    uint32_t *counts;
    uint2 *cand;
    uint32_t cap;
};

constexpr int lowp_lds_bytes(int nwaves) { return 3 * MF_STAGE_BYTES + nwaves * 512 + MF_EQ_BYTES; }

template <int LK, int KSTEPS, int MODE, int RT, int NWAVES, int NQW, int MINW = 1>
__global__ __launch_bounds__(NWAVES * 64, MINW) void k_mfma_filter_lowp(LowpParams P) {
    using Ops = LowpOps<LK>;
    using acc_t = typename Ops::acc_t;
    constexpr int NS = 3;
    constexpr int MT = RT / 16;
    constexpr int SEG = MF_STAGE_BYTES / RT;    // bytes per row per stage: 1024 / 512 / 256
    constexpr int KSUB = SEG / 64;              // k-steps (64 B of row each) per stage
    static_assert(KSTEPS % KSUB == 0, "row bytes must be a multiple of the stage segment");
    constexpr int KCH = KSTEPS / KSUB;
    static_assert(KCH >= 2, "a tile must span at least two ring slots");
    constexpr int IPW = 16 / NWAVES;            // DMA instructions per wave per stage
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15;
    const int kq = lane >> 4;
    const int qtile = blockIdx.y;

    u32x4_t qf[NQW][KSTEPS];
    {
        const u32x4_t *src = reinterpret_cast<const u32x4_t *>(P.qfrag) + ((size_t)((qtile * NWAVES + wave) * NQW) * KSTEPS) * 64 + lane;
#pragma unroll
        for (int nt = 0; nt < NQW; nt++)
#pragma unroll
            for (int s = 0; s < KSTEPS; s++) qf[nt][s] = src[(size_t)(nt * KSTEPS + s) * 64];
    }
    int qidx[NQW];
    uint32_t qaux[NQW];
    float tau[NQW];
#pragma unroll
    for (int nt = 0; nt < NQW; nt++) {
        qidx[nt] = qtile * (NWAVES * 16 * NQW) + wave * (16 * NQW) + nt * 16 + m16;
        qaux[nt] = P.qaux[qidx[nt]];
        tau[nt] = (MODE == MF_FILTER) ? P.tau[qidx[nt]] : 0.f;
    }
    // pin the ordinary loads before the first DMA (see k_mfma_filter)
#pragma unroll
    for (int nt = 0; nt < NQW; nt++) {
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) asm volatile("" : "+v"(qf[nt][s]));
        asm volatile("" : "+v"(qaux[nt]), "+v"(tau[nt]));
    }

    // staging geometry: instruction g = wave*IPW + t fills LDS bytes [1024 g, 1024 g + 1024)
    uint32_t st_row[IPW], st_off[IPW];
#pragma unroll
    for (int t = 0; t < IPW; t++) {
        const uint32_t L = 1024u * (uint32_t)(IPW * wave + t) + 16u * (uint32_t)lane;
        const uint32_t row = L / SEG, slot = (L % SEG) / 16;
        st_row[t] = row;
        st_off[t] = (slot / 16) * 256 + (((slot % 16) ^ (row & 15)) * 16);
    }
    const uint32_t lds_stage_wave_off = (uint32_t)(wave * IPW * 1024);
    char *aux_lds = lds + NS * MF_STAGE_BYTES + wave * 512;
    uint32_t *eq_n = reinterpret_cast<uint32_t *>(lds + NS * MF_STAGE_BYTES + NWAVES * 512);
    uint4 *eq = reinterpret_cast<uint4 *>(lds + NS * MF_STAGE_BYTES + NWAVES * 512 + 16);
    const uint32_t eq_n_off = mf_lds_offset(eq_n), eq_off = mf_lds_offset(eq);
    if (MODE == MF_FILTER && tid == 0) *eq_n = 0;

    const uint32_t step = gridDim.x;
    auto tile_row0 = [&](uint32_t t) -> uint32_t { return (P.tile_first + t * P.tile_step) * RT; };
    const char *rp_cur[IPW], *rp_nxt[IPW];
    const uint32_t *ap_cur, *ap_nxt;
    auto make_ptrs = [&](uint32_t t, const char *(&rp)[IPW], const uint32_t *&ap) {
        uint32_t tt = t < P.n_tiles ? t : P.n_tiles - 1;
        const uint32_t r0 = tile_row0(tt);
        const uint32_t sidx = __builtin_amdgcn_readfirstlane(r0 >> P.slab_shift);
        const char *sbase = P.slabs[sidx];
        const uint32_t *abase = P.aux_slabs[sidx];
#pragma unroll
        for (int i = 0; i < IPW; i++) {
            uint32_t row = r0 + st_row[i];
            if (row >= P.n_rows) row = P.n_rows - 1;
            rp[i] = sbase + (size_t)(row & P.slab_mask) * P.row_stride + st_off[i];
        }
        uint32_t arow = r0 + lane;
        if (arow >= P.n_rows) arow = P.n_rows - 1;
        ap = abase + (arow & P.slab_mask);
    };
    auto issue = [&](const char *const (&rp)[IPW], const uint32_t *ap, int kc, uint32_t slot, bool with_aux,
                     uint32_t parity) {
        const uint32_t base = slot * MF_STAGE_BYTES + lds_stage_wave_off;
#pragma unroll
        for (int i = 0; i < IPW; i++) glds16<2>(rp[i] + (size_t)kc * SEG, base + i * 1024, lds);
        if (with_aux) glds4(ap, parity * 256, aux_lds);
    };

    uint32_t tile = blockIdx.x;
    make_ptrs(tile, rp_cur, ap_cur);
    make_ptrs(tile + step, rp_nxt, ap_nxt);
    uint32_t slot_c = 0, parity = 0;
    issue(rp_cur, ap_cur, 0, 0, true, 0);
    issue(rp_cur, ap_cur, 1, 1, false, 0);

    for (; tile < P.n_tiles; tile += step) {
        acc_t acc[MT][NQW];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NQW; nt++) acc[mt][nt] = acc_t{0, 0, 0, 0};

#pragma unroll
        for (int c = 0; c < KCH; c++) {
            // unit c landed; unit c+1 (IPW loads, +1 aux load if it opens the next tile) may stay in flight
            if (IPW == 1) {
                if (c + 1 == KCH) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            } else if (IPW == 2) {
                if (c + 1 == KCH) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                if (c + 1 == KCH) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (MODE == MF_FILTER && c == 0) {
                if (*eq_n >= MF_EQ_CAP / 2) mf_flush_queue<NWAVES * 64>(eq_n, eq, P.counts, P.cand, P.cap);
            }
            {
                uint32_t slot_p = slot_c + 2;
                if (slot_p >= NS) slot_p -= NS;
                if (c + 2 < KCH) issue(rp_cur, ap_cur, c + 2, slot_p, false, 0);
                else issue(rp_nxt, ap_nxt, c + 2 - KCH, slot_p, (c + 2 - KCH) == 0, parity ^ 1u);
            }
            const char *sbase = lds + slot_c * MF_STAGE_BYTES;
            // LDS reads are issued in groups of PF fragments ahead of the MFMAs that consume them, so the
            // ds_read latency overlaps the matrix pipe instead of serialising with it (hipcc otherwise
            // emits read -> wait -> mfma per fragment)
            constexpr int NFRAG = KSUB * MT;
            constexpr int PF = NFRAG < 8 ? NFRAG : 8;
#pragma unroll
            for (int g = 0; g < NFRAG; g += PF) {
                u32x4_t afr[PF];
#pragma unroll
                for (int f = 0; f < PF; f++) {
                    const int j = (g + f) / MT, mt = (g + f) % MT;
                    const char *rowp = sbase + (mt * 16 + m16) * SEG + (j / 4) * 256;
                    const int p = (4 * (j % 4) + kq) ^ m16;
                    afr[f] = *reinterpret_cast<const u32x4_t *>(rowp + p * 16);
                }
#pragma unroll
                for (int f = 0; f < PF; f++) {
                    const int j = (g + f) / MT, mt = (g + f) % MT;
#pragma unroll
                    for (int nt = 0; nt < NQW; nt++) acc[mt][nt] = Ops::mma(afr[f], qf[nt][c * KSUB + j], acc[mt][nt]);
                }
            }
            slot_c = slot_c + 1 == NS ? 0 : slot_c + 1;
        }

        // ---- epilogue ----
        const uint32_t r0 = tile_row0(tile);
        const uint32_t *aux = reinterpret_cast<const uint32_t *>(aux_lds + parity * 256);
        bool emitted = false;
        float tmin[NQW];
#pragma unroll
        for (int nt = 0; nt < NQW; nt++) tmin[nt] = INFINITY;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const uint4 a4 = *reinterpret_cast<const uint4 *>(aux + mt * 16 + kq * 4);
            const uint32_t av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t row = r0 + mt * 16 + kq * 4 + i;
#pragma unroll
                for (int nt = 0; nt < NQW; nt++) {
                    float low, up;
                    if (LK == LP_I8) {
                        const int dot = (int)acc[mt][nt][i];
                        float s;
                        if (P.epi == LE_I8_L2) s = (float)((int)av[i] + (int)qaux[nt] - 2 * dot);
                        else if (P.epi == LE_I8_IP) s = (float)(1 - dot);
                        else s = __fsub_rn(1.0f, __fdiv_rn((float)dot, __fmul_rn(__uint_as_float(av[i]), __uint_as_float(qaux[nt]))));
                        low = up = s;
                    } else {
                        const float dot = (float)acc[mt][nt][i];
                        const float ssum = __uint_as_float(av[i]) + __uint_as_float(qaux[nt]);
                        const float a = (P.epi == LE_FP_L2) ? (ssum - 2.0f * dot) : (1.0f - dot);
                        const float E = P.cE * ssum + P.absE;
                        low = a - E;
                        up = a + E;
                    }
                    if (MODE == MF_PROBE) {
                        if (row < P.n_rows && up < tmin[nt]) tmin[nt] = up;
                    } else if (row < P.n_rows && low <= tau[nt]) {
                        const uint32_t pos = mf_queue_reserve(eq_n_off);
                        if (pos < MF_EQ_CAP) {
                            mf_queue_write(eq_off + pos * 16, row, (uint32_t)qidx[nt], __float_as_uint(low));
                        } else {
                            uint32_t s = atomicAdd(&P.counts[qidx[nt]], 1u);
                            if (s < P.cap) P.cand[(size_t)qidx[nt] * P.cap + s] = make_uint2(row, __float_as_uint(low));
                            emitted = true;
                        }
                    }
                }
            }
        }
        if (MODE == MF_PROBE) {
#pragma unroll
            for (int nt = 0; nt < NQW; nt++) {
                float v = tmin[nt];
                v = fminf(v, __shfl_xor(v, 16));
                v = fminf(v, __shfl_xor(v, 32));
                if (kq == 0) P.tilemin[(size_t)qidx[nt] * P.tilemin_stride + tile] = v;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (__any(emitted)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < IPW; i++) rp_cur[i] = rp_nxt[i];
        ap_cur = ap_nxt;
        make_ptrs(tile + 2 * step, rp_nxt, ap_nxt);
        parity ^= 1u;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (MODE == MF_FILTER) {
        __builtin_amdgcn_s_barrier();
        mf_flush_queue<NWAVES * 64>(eq_n, eq, P.counts, P.cand, P.cap);
    }
}

// ---- per-row aux values ----
// bf16 / fp16: |x|^2 in double -> float.  kind: 2 = bf16, 3 = fp16 (VSGPU type codes)
__global__ __launch_bounds__(256) void k_row_norms_h16(const char *rows, uint32_t row_stride, uint32_t dim, uint32_t n,
                                                       int kind, uint32_t *out) {
    const int lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const uint16_t *p = reinterpret_cast<const uint16_t *>(rows + (size_t)row * row_stride);
    double s = 0.0;
    for (uint32_t i = lane; i < dim; i += 64) {
        const uint16_t h = p[i];
        float f = (kind == 2) ? __uint_as_float((uint32_t)h << 16) : (float)__builtin_bit_cast(_Float16, h);
        s += (double)f * (double)f;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[row] = __float_as_uint((float)s);
}
// int8: sum x^2 as int32 (mode 0) or the float norm stored after the elements (mode 1, Cosine rows)
__global__ __launch_bounds__(256) void k_row_aux_i8(const char *rows, uint32_t row_stride, uint32_t dim, uint32_t n,
                                                    int mode, uint32_t *out) {
    const int lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const char *p = rows + (size_t)row * row_stride;
    if (mode == 1) {
        if (lane == 0) {
            const unsigned char *np = reinterpret_cast<const unsigned char *>(p + dim);
            out[row] = (uint32_t)np[0] | ((uint32_t)np[1] << 8) | ((uint32_t)np[2] << 16) | ((uint32_t)np[3] << 24);
        }
        return;
    }
    int s = 0;
    for (uint32_t i = lane; i < dim; i += 64) {
        const int v = (int)*reinterpret_cast<const int8_t *>(p + i);
        s += v * v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[row] = (uint32_t)s;
}

}  // namespace vsg
