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

#include <type_traits>

#include "mfma_kernels.hpp"

namespace vsg {

enum LowpKind { LP_BF16 = 0, LP_F16 = 1, LP_I8 = 2, LP_U8 = 3, LP_SQ8 = 4, LP_U8C = 5 };
enum LowpEpi { LE_FP_L2 = 0, LE_FP_IP = 1, LE_I8_L2 = 2, LE_I8_IP = 3, LE_I8_COS = 4, LE_U8_IP = 5 };

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
// uint8 rows ride the signed-int8 MFMA: x ^ 0x80 is x - 128 as int8 (applied to the A fragments after the LDS
// read and to the query fragments on the host), and
//   L2:  sum (x-q)^2 = sum x'^2 + sum q'^2 - 2 sum x'q'              (shift invariant; aux = sum x'^2)
//   IP:  sum x q     = sum x'q' + 128 sum x' + (128 sum q' + 128^2 d)  (aux = sum x', per-query constant)
// are exact integers, so the reference's epilogues (L2.cpp:164-174, IP.cpp:273-286) apply unchanged.
template <> struct LowpOps<LP_U8> {
    using acc_t = i32x4_t;
    __device__ static inline acc_t mma(u32x4_t a, u32x4_t b, acc_t c) {
        return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_t, a), __builtin_bit_cast(i32x4_t, b), c, 0, 0, 0);
    }
};
// SQ8 storage x FP32 query (types/sq8.h; IP.cpp:34-80, L2.cpp:30-45): the codes ride the signed MFMA re-centred by 128
// like uint8 rows; the fp32 query is quantised to ONE int8 piece per element on the host, y_i = s Y_i + e_i with
// |Y_i| <= 127, so with c'_i = c_i - 128
//     sum c_i y_i = s (D + 128 sum Y) + 128 sum e_i + sum c'_i e_i,      D = sum c'_i Y_i exact in int32,
// and |sum c'_i e_i| <= |c'|_2 |e|_2 (Cauchy-Schwarz).  The kernel turns that into bounds on the reference's score from
// per-row {min, delta, sum_squares, |c'|_2} (16 B of aux per row) and per-query {s, 128 sum Y, y_sum, y_sum_squares,
// 128 sum e, |e|_2, Wref} (Wref: the reference's own fp32 rounding of the dot product); survivors are re-scored by
// k_exact_pairs in the reference's lane order.
template <> struct LowpOps<LP_SQ8> {
    using acc_t = i32x4_t;
    __device__ static inline acc_t mma(u32x4_t a, u32x4_t b, acc_t c) {
        return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_t, a), __builtin_bit_cast(i32x4_t, b), c, 0, 0, 0);
    }
};
// uint8 Cosine: the re-centred dot needs TWO per-row values, sum x' for  sum x q = sum x'q' + 128 sum x' + (128 sum q' +
// 128^2 d)  and the stored float norm for the reference's epilogue 1 - float(dot) / (norm_x norm_q) (IP.cpp:264-271 twin for
// uint8) -- the 16-byte aux records of the SQ8 path carry both.  Exact integer dot, exact score: no re-rank.
template <> struct LowpOps<LP_U8C> {
    using acc_t = i32x4_t;
    __device__ static inline acc_t mma(u32x4_t a, u32x4_t b, acc_t c) {
        return __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_t, a), __builtin_bit_cast(i32x4_t, b), c, 0, 0, 0);
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
    uint32_t tile_run_shift;                   // ... probe: in runs of 2^shift consecutive tiles (see MfmaParams)
    // SQ8 filter, per-value screen: {max delta, max |min|, max sum_squares} over the table (float bits, k_row_aux_sq8) and
    // the largest |code dot + K| / |c - 128|_2 any row of this width can have
    const uint32_t *sq8_max;             // (int8 / uint8 tables: {min, max} of the rows' aux values as ints, k_row_aux_i8)
    float sq8_fmax, sq8_ncmax;
    const uint4 *qfrag;                  // [q_tile][wave][NQW][KSTEPS][lane] 16-B B-operand fragments
    const uint32_t *qaux;                // per query: float |q|^2 | int32 sum q^2 | float norm
    const float *qmeta;                  // LP_SQ8: [queries][8] = {s, bits(int 128 sum Y), y_sum, y_sum_squares, Wref, 128 sum e, |e|_2, 0}
                                         // LP_U8C: {norm_q, bits(int 128 sum q' + 16384 dim), 0 ...}
    int epi;
    float cE, absE;
    float *tilemin;                      // MF_PROBE: [queries][tilemin_stride]
    size_t tilemin_stride;
    const float *tau;                    // MF_FILTER
    uint32_t *counts;
    uint2 *cand;
    uint32_t cap;
    int dbg;                             // diagnosis switches (vsgpu option lowp_dbg), 0 in production
    // pair_map != 0: 1-D grid of 2*G workgroups; two query tiles walk the same row tiles from the same XCD
    // (ids 8 apart are dispatched back to back onto one XCD), the second reader is then served by that XCD's L2
    int pair_map;
};

// Ring geometry.  NS slots of STAGE bytes (16 KiB unless stated); a slot ("unit") holds RT rows x SEG bytes, a tile is KCH units.  NS-1
// units are requested ahead of the one being consumed, which reaches TA tiles ahead, so TA+1 per-tile aux
// buffers (256 B per wave each) are live at once.
constexpr int lowp_kch(int ksteps, int rt, int stage) { return ksteps / ((stage / rt) / 64); }
constexpr int lowp_ta(int ksteps, int rt, int ns, int stage) {
    return (lowp_kch(ksteps, rt, stage) - 1 + (ns - 1)) / lowp_kch(ksteps, rt, stage);
}
constexpr int LOWP_WQ_CAP = 64;  // SKEW: records per wave-private queue (a 16-byte header holds the fill count)
// MF_PROBE: per (tile, query) minima wait in LDS and leave in batches of LOWP_PM_TILES tiles per wave -- a global store per
// tile shares the VM counter with the ring and costs a full drain (s_waitcnt vmcnt(0)) per tile
constexpr int LOWP_PM_TILES = 32;
constexpr int lowp_pm_bytes(int nwaves, int nqw) { return LOWP_PM_TILES * nwaves * 16 * nqw * 4; }
constexpr int lowp_lds_bytes(int nwaves, int ksteps, int rt, int ns, int stage = MF_STAGE_BYTES, bool skew = false, int probe_nqw = 0) {
    return ns * stage + nwaves * 256 * (lowp_ta(ksteps, rt, ns, stage) + 1) +
           (skew ? nwaves * (16 + LOWP_WQ_CAP * 16) : MF_EQ_BYTES) + (probe_nqw ? lowp_pm_bytes(nwaves, probe_nqw) : 0);
}

// s_waitcnt needs an immediate: after unrolling, n is a constant and the switch folds to one instruction
__device__ static inline void lowp_wait_vmcnt(int n) {
    switch (n) {
#define VSG_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    VSG_W(0) VSG_W(1) VSG_W(2) VSG_W(3) VSG_W(4) VSG_W(5) VSG_W(6) VSG_W(7) VSG_W(8) VSG_W(9) VSG_W(10) VSG_W(11)
    VSG_W(12) VSG_W(13) VSG_W(14) VSG_W(15) VSG_W(16) VSG_W(17) VSG_W(18) VSG_W(19) VSG_W(20) VSG_W(21) VSG_W(22)
    VSG_W(23) VSG_W(24) VSG_W(25) VSG_W(26) VSG_W(27) VSG_W(28) VSG_W(29) VSG_W(30) VSG_W(31) VSG_W(32) VSG_W(33)
    VSG_W(34) VSG_W(35) VSG_W(36) VSG_W(37) VSG_W(38) VSG_W(39) VSG_W(40)
#undef VSG_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// s_waitcnt lgkmcnt(n) that also orders the consumer of `x` (an inline-asm ds_read's destination) behind the wait
template <typename T> __device__ static inline void lowp_wait_lgkmcnt(int n, T &x) {
    switch (n) {
#define VSG_W(N) case N: asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(x)); break;
    VSG_W(1) VSG_W(2) VSG_W(3) VSG_W(4) VSG_W(5) VSG_W(6) VSG_W(7) VSG_W(8) VSG_W(9) VSG_W(10) VSG_W(11) VSG_W(12) VSG_W(13) VSG_W(14) VSG_W(15)
#undef VSG_W
    default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x)); break;
    }
}

#ifndef LOWP_PF
#define LOWP_PF 4
#endif

template <int LK, int KSTEPS, int MODE, int RT, int NWAVES, int NQW, int MINW = 1, int NS = 3, int STAGE = MF_STAGE_BYTES,
          bool SKEW = false, int DIST = 0, int DLATE = 0, bool DIAG = false, int ISS = 0>
__global__ __launch_bounds__(NWAVES * 64, MINW) void k_mfma_filter_lowp(LowpParams P) {
    using Ops = LowpOps<LK>;
    using acc_t = typename Ops::acc_t;
    constexpr bool U8C = (LK == LP_U8C);
    constexpr bool SQ8 = (LK == LP_SQ8) || U8C;   // (the 16-byte aux record path; the name stuck)
    constexpr int AUXBUF = SQ8 ? 1024 : 256;   // bytes of per-tile aux values: 4 B per row, 16 B per row for SQ8 / uint8 Cosine
    // UADDR: request addresses as a wave-uniform 64-bit base (SGPRs) + one constant 32-bit offset per lane and piece (the
    // scalar-base form of global_load_lds) instead of a 64-bit address per lane and piece.  Rows past the table's end are not
    // clamped: a tile never leaves its slab, whose allocation is whole, and the epilogues mask such rows (nvalid).
    // Round 3 used it for SQ8 only; round 4, same-box A/B on config 4 (bf16 768, 128 queries): scan kernel 3.33 -> 3.26 ms, so every
    // kind gets it (-DLOWP_UADDR_ALL=0 restores the per-lane addresses).
#ifndef LOWP_UADDR_ALL
#define LOWP_UADDR_ALL 1
#endif
    constexpr bool UADDR = (LK == LP_SQ8) || (LOWP_UADDR_ALL && !SKEW);
    static_assert(!SQ8 || (!SKEW && NQW == 1 && RT * 16 <= AUXBUF && NWAVES * 256 >= AUXBUF), "SQ8 aux geometry");
    static_assert(LK != LP_SQ8 || RT == 64, "SQ8 aux arrays are laid out per 64 rows (k_row_aux_sq8)");
    // units requested ahead.  DIST = NS-2 leaves one slot of slack: the slot refilled after a barrier was last read
    // a whole unit earlier, so the plain s_barrier is enough and hipcc may keep pipelining LDS reads across it
    constexpr int D = DIST ? DIST : NS - 1;
    constexpr bool SLACK = (NS - D) >= 2;
    constexpr int MT = RT / 16;
    constexpr int SEG = STAGE / RT;    // bytes per row per stage: 1024 / 512 / 256
    constexpr int KSUB = SEG / 64;              // k-steps (64 B of row each) per stage
    static_assert(KSTEPS % KSUB == 0, "row bytes must be a multiple of the stage segment");
    constexpr int KCH = KSTEPS / KSUB;
    // ISS > 0: only waves 0 .. ISS-1 (two per SIMD when ISS = NWAVES/2) request rows, IPW pieces each; the others go
    // from the barrier straight to their fragment reads, so their MFMAs cover the issuers' LDS-DMA issue time
    constexpr int NISS = ISS ? ISS : NWAVES;
    static_assert(!ISS || (!SKEW && ISS < NWAVES), "issuer subset");
    constexpr int IPW = (STAGE / 1024) / NISS;  // DMA instructions (1 KiB each) per requesting wave per stage
    static_assert(IPW >= 1 && SEG * RT == STAGE && KSUB >= 1, "stage geometry");
    constexpr int TA = (KCH - 1 + D) / KCH;     // tiles ahead reached by the prefetch
    constexpr int NAUX = TA + 1;
    static_assert((D - 1) * IPW + D <= 40, "vmcnt immediate table");
    static_assert(!SKEW || (MODE == MF_FILTER && D >= 2 && (NWAVES == 8 || NWAVES == 16)), "phase-skewed variant");
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    // the diagnosis switches and the paired mapping are compiled in only where asked for: as run-time flags they
    // put a branch around every DMA piece of the main loop
    const int dbg = DIAG ? P.dbg : 0;
    const int pair_map = DIAG ? P.pair_map : 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = SKEW ? ((wave >> 2) & 1) : 0;  // SKEW: waves w and w+4 share a SIMD, one of each half
    const int m16 = lane & 15;
    const int kq = lane >> 4;
    const int qtile = pair_map ? (int)((blockIdx.x >> 3) & 1u) : (int)blockIdx.y;

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
    // SQ8 / uint8 Cosine: the per-query constants of the epilogue, loaded once (a global load inside the tile loop would put
    // a vmcnt(0) -- a drain of the DMA ring -- into every tile)
    float qm_r[7] = {0, 0, 0, 0, 0, 0, 0};
    if (SQ8) {
        const float *qm = P.qmeta + (size_t)qidx[0] * 8;
#pragma unroll
        for (int j = 0; j < 7; j++) qm_r[j] = qm[j];
#pragma unroll
        for (int j = 0; j < 7; j++) asm volatile("" : "+v"(qm_r[j]));
    }
    // SQ8: per-query constants, kept as few registers as the screen allows (the kernel runs at 128 VGPRs, two workgroups per CU):
    // with g = 2 (L2) or 1 (IP)   sq_a = {-g s, -g 128 sum e}   sq_b = {-g y_sum, -g |e|_2}   sq_c = {y_sq | shift, Wref}
    // (exact scalings, so the per-value pass gets s, y_sum ... back from them), sq_T = the screen's threshold, sq_K = 128 sum Y.
    f32x2_t sq_a = {0, 0}, sq_b = {0, 0}, sq_c = {0, 0};
    float sq_T = 0.f;
    int sq_K = 0;
    if (LK == LP_SQ8) {
        const bool l2 = P.epi == LE_FP_L2;
        const float G = l2 ? 2.0f : 1.0f;
        const float qs = qm_r[0], ysum = qm_r[2], ysq = qm_r[3], Wref = qm_r[4], ce = qm_r[5], ne = qm_r[6];
        sq_K = (int)__float_as_uint(qm_r[1]);
        sq_a = f32x2_t{-G * qs, -G * ce};
        sq_b = f32x2_t{-G * ysum, -G * ne};
        sq_c = f32x2_t{ysq, Wref};
        if (MODE == MF_FILTER) {
            const float dl_hi = __uint_as_float(P.sq8_max[0]), mn_hi = __uint_as_float(P.sq8_max[1]), xsq_hi = __uint_as_float(P.sq8_max[2]);
            // every magnitude a rounding of the screen, of `bounds` or of the reference can be relative to, from table-wide maxima
            const float tmax = fabsf(qs) * P.sq8_fmax + fabsf(ce);
            const float mag = G * (mn_hi * fabsf(ysum) + dl_hi * (tmax + P.sq8_ncmax * ne)) + (l2 ? xsq_hi : 1.0f) + fabsf(ysq) + fabsf(tau[0]);
            const float slack = (G * dl_hi * Wref) * 1.0001f + (192.0f / 16777216.0f) * mag;
            sq_T = l2 ? (tau[0] - ysq) + slack : tau[0] + slack;
        }
        asm volatile("" : "+v"(sq_a), "+v"(sq_b), "+v"(sq_c), "+v"(sq_T), "+v"(sq_K));
    }
    // pin the ordinary loads before the first DMA (see k_mfma_filter)
#pragma unroll
    for (int nt = 0; nt < NQW; nt++) {
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
            // 256 registers of query fragments (4 column blocks x 16 k-steps): the upper half is parked in AGPRs, which
            // the MFMA reads directly; left to hipcc they are spilled there and copied back in front of every MFMA
            if (NQW * KSTEPS * 4 > 192 && nt >= NQW / 2) asm volatile("" : "+a"(qf[nt][s]));
            else asm volatile("" : "+v"(qf[nt][s]));
        }
        asm volatile("" : "+v"(qaux[nt]), "+v"(tau[nt]));
    }
    // int8 Cosine: the exact score needs an IEEE divide per (row, query).  1 - dot/(nx*nq) <= tau  <=>
    // dot >= (1 - tau) * nq * nx, so a row is first screened with one multiply against cosq = ((1 - tau) - m) * nq,
    // m covering every rounding on either side (each is below 2^-21 relative to |1 - tau| <= 3); only rows that
    // pass -- a handful per query -- get the exact score and the exact test.
    float cosq[NQW];
#pragma unroll
    for (int nt = 0; nt < NQW; nt++) {
        const float omt = 1.0f - tau[nt];
        cosq[nt] = (omt - 1e-5f * (1.0f + fabsf(omt))) * __uint_as_float(qaux[nt]);
    }

    // bf16 / fp16 rows, FILTER: whether ANY value of a tile survives is decided with two operations per value instead of six.
    // The exact test keeps a value unless  a - E > tau,  a = (n + nq2) - 2 dot | 1 - dot,  E = cE (n + nq2) + absE  (n = the
    // row's |x|^2).  Per query  fpq = tau + absE - nq2 (1 - cE)  |  (tau - 1) + cE nq2 + absE,  per row  rowt = n (1 - cE) | n cE,
    // and the value is dropped when  rowt - 2 dot > fpq  |  -dot - rowt > fpq.  That form rounds differently from the exact
    // one, so it is made strictly more permissive -- 2^-20 (16 ulp) of every magnitude involved on the keeping side -- and a
    // tile it lets through runs the exact test as before: replies cannot change, only how rarely the slow path is entered.
    float fpq[NQW];
    float fp_rowc = 0.f;
    if (MODE == MF_FILTER && !SQ8 && (LK == LP_BF16 || LK == LP_F16)) {
        const bool l2q = P.epi == LE_FP_L2;
        const float slk = 9.5367431640625e-07f;   // 2^-20
        fp_rowc = l2q ? (1.0f - P.cE) - slk : P.cE + slk;
#pragma unroll
        for (int nt = 0; nt < NQW; nt++) {
            const float nq2 = __uint_as_float(qaux[nt]);
            const float tq = l2q ? (tau[nt] + P.absE) - nq2 * (1.0f - P.cE) : (tau[nt] - 1.0f) + (P.cE * nq2 + P.absE);
            fpq[nt] = tau[nt] == -INFINITY ? -INFINITY : tq + slk * (fabsf(tau[nt]) + nq2 + 2.0f);   // (padding queries: tau = -inf)
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < NQW; nt++) fpq[nt] = 0.f;
    }

    // staging geometry: instruction g = wave*IPW + t fills LDS bytes [1024 g, 1024 g + 1024)
    uint32_t st_row[IPW], st_off[IPW];
    uint32_t st_row0[IPW], st_lane[IPW];   // UADDR: the piece's first row (uniform) and the lane's byte offset from that row's start
#pragma unroll
    for (int t = 0; t < IPW; t++) {
        const uint32_t L = 1024u * (uint32_t)(IPW * wave + t) + 16u * (uint32_t)lane;
        const uint32_t row = L / SEG, slot = (L % SEG) / 16;
        st_row[t] = row;
        st_off[t] = (slot / 16) * 256 + (((slot % 16) ^ (row & 15)) * 16);
        st_row0[t] = (1024u * (uint32_t)(IPW * wave + t)) / SEG;
        st_lane[t] = (row - st_row0[t]) * P.row_stride + st_off[t];
    }
    const uint32_t lds_stage_wave_off = (uint32_t)(wave * IPW * 1024);
    const bool issuer = !ISS || wave < NISS;
    // per-tile aux values: one copy per workgroup, requested by wave 0 (the unit barrier publishes it) -- every LDS-DMA
    // piece costs its wave hundreds of issue cycles, and 16 waves each fetching the same 256 bytes was a third of all
    // pieces of the int8 kernel.  (SKEW keeps a private copy per wave: its halves pass different barriers.)
    const bool aux_loader = SKEW || wave == 0;
    char *aux_lds = lds + NS * STAGE + (SKEW ? wave : 0) * (AUXBUF * NAUX);
    uint32_t *eq_n = reinterpret_cast<uint32_t *>(lds + NS * STAGE + NWAVES * 256 * NAUX);
    uint4 *eq = reinterpret_cast<uint4 *>(lds + NS * STAGE + NWAVES * 256 * NAUX + 16);
    const uint32_t eq_n_off = mf_lds_offset(eq_n), eq_off = mf_lds_offset(eq), aux_lds_off = mf_lds_offset(aux_lds);
    // SKEW: wave-private queue [fill count, pad][LOWP_WQ_CAP records] in place of the workgroup queue
    const uint32_t wq_cnt_off = mf_lds_offset(lds + NS * STAGE + NWAVES * 256 * NAUX) + (uint32_t)wave * (16 + LOWP_WQ_CAP * 16);
    const uint32_t q_cnt_off = SKEW ? wq_cnt_off : eq_n_off, q_rec_off = SKEW ? wq_cnt_off + 16 : eq_off;
    constexpr uint32_t Q_CAP = SKEW ? LOWP_WQ_CAP : MF_EQ_CAP;
    if (SKEW) {
        const uint32_t zero = 0;
        if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(wq_cnt_off), "v"(zero) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (MODE == MF_FILTER && tid == 0) {
        *eq_n = 0;
    }
    // SKEW: a wave empties its own queue, no barrier involved
    auto drain_wave_queue = [&]() {
        uint32_t n;
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(n) : "v"(wq_cnt_off) : "memory");
        n = min(n, (uint32_t)LOWP_WQ_CAP);
        if ((uint32_t)lane < n) {
            u32x4_t r;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(wq_cnt_off + 16u + (uint32_t)lane * 16u) : "memory");
            const uint32_t s = atomicAdd(&P.counts[r[1]], 1u);
            if (s < P.cap) P.cand[(size_t)r[1] * P.cap + s] = make_uint2(r[0], r[2]);
        }
        const uint32_t zero = 0;
        if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(wq_cnt_off), "v"(zero) : "memory");
        // stores/atomics share the VM counter with the staged loads: drain so the counted waits stay exact
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };

    const uint32_t step = pair_map ? gridDim.x / 2 : gridDim.x;
    auto tile_row0 = [&](uint32_t t) -> uint32_t {
        return (P.tile_first + (t >> P.tile_run_shift) * (P.tile_step << P.tile_run_shift) + (t & ((1u << P.tile_run_shift) - 1u))) * RT;
    };
    // Requests are issued strictly in unit order, so only the frontier tile's addresses are kept
    const char *rp_f[IPW];
    const uint32_t *ap_f;
    // slab base pointers are scalar loads (hundreds of cycles each, and they drain lgkmcnt): reload only when
    // a tile crosses into another slab
    uint32_t cur_slab = 0xFFFFFFFFu;
    uint64_t cur_sbase = 0, cur_abase = 0;
    auto make_ptrs = [&](uint32_t t, const char *(&rp)[IPW], const uint32_t *&ap) {
        if (!issuer) return;
        uint32_t tt = t < P.n_tiles ? t : P.n_tiles - 1;
        const uint32_t r0 = UADDR ? (uint32_t)__builtin_amdgcn_readfirstlane(tile_row0(tt)) : tile_row0(tt);
        const uint32_t sidx = __builtin_amdgcn_readfirstlane(r0 >> P.slab_shift);
        if (sidx != cur_slab) {
            // asm scalar loads: left to hipcc these become vector loads, and the vmcnt(0) it then needs on the
            // join path drains the whole DMA ring at every tile
            cur_slab = sidx;
            const char *const *sp = P.slabs + sidx;
            const uint32_t *const *axp = P.aux_slabs + sidx;
            asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(cur_sbase), "=&s"(cur_abase)
                         : "s"(sp), "s"(axp)
                         : "memory");
        }
        const char *sbase = reinterpret_cast<const char *>(cur_sbase);
        const uint32_t *abase = reinterpret_cast<const uint32_t *>(cur_abase);
        if (UADDR) {
            // rp[0] = the tile's first row (uniform); the pieces add their row and lane offsets when they are issued
            rp[0] = sbase + (size_t)(r0 & P.slab_mask) * P.row_stride;
            ap = abase + (size_t)(r0 & P.slab_mask) * (SQ8 ? 4 : 1);
            return;
        }
#pragma unroll
        for (int i = 0; i < IPW; i++) {
            uint32_t row = r0 + st_row[i];
            if (row >= P.n_rows) row = P.n_rows - 1;
            rp[i] = sbase + (size_t)(row & P.slab_mask) * P.row_stride + st_off[i];
        }
        uint32_t arow = r0 + lane;
        // (LP_SQ8: lane l fetches 16 bytes of the tile's four aux arrays, not row l's record: no clamp -- aux slabs are whole groups)
        if (LK != LP_SQ8 && arow >= P.n_rows) arow = P.n_rows - 1;
        ap = abase + (size_t)(arow & P.slab_mask) * (SQ8 ? 4 : 1);
    };
    // SQ8 rows (dim + 12 / 16 bytes: never a multiple of the 128-byte line at the compiled widths) are requested in column-block
    // segments that START AND END inside lines: the neighbouring segment -- another stage, about a unit later -- touches the same line
    // again.  Marked non-temporal the line is gone by then and crosses HBM twice (round 5, rocprofv3 PMC on the SQ8 filter, 4 M x 784 B:
    // 1.26 x the algorithmic bytes); with the default policy L2 keeps it (1.03 x).  A compile-time property of the kind on purpose: the
    // same test at run time (any stride % 128 != 0) put a branch around every request of the bf16 filter and cost config 4 5 %.
    constexpr bool keep_lines = (LK == LP_SQ8);
    auto issue = [&](const char *const (&rpt)[IPW], const uint32_t *apt, int kc, uint32_t slot, bool with_aux,
                     uint32_t abuf_i) {
        if (!issuer) return;
        const uint32_t base = slot * STAGE + lds_stage_wave_off;
        if (!(dbg & 4)) {
#pragma unroll
            for (int i = 0; i < IPW; i++) {
                if (UADDR) {
                    const uint64_t pb = reinterpret_cast<uint64_t>(rpt[0]) + (uint64_t)st_row0[i] * P.row_stride + (uint64_t)kc * SEG;
                    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(pb >> 32)) << 32) |
                                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pb);   // (uniform already: keeps it in SGPRs)
                    uint32_t lo32 = st_lane[i];
                    asm volatile("" : "+v"(lo32));   // (keeps the zero-extension in this block: hoisted, the scalar-base form is not selected)
                    if (pair_map || keep_lines) glds16<0>(reinterpret_cast<const char *>(pu) + lo32, base + i * 1024, lds);
                    else glds16<2>(reinterpret_cast<const char *>(pu) + lo32, base + i * 1024, lds);
                    continue;
                }
                // paired query tiles want the row to stay in L2 for the partner: default cache policy there
                if (pair_map || keep_lines) glds16<0>(rpt[i] + (size_t)kc * SEG, base + i * 1024, lds);
                else glds16<2>(rpt[i] + (size_t)kc * SEG, base + i * 1024, lds);
            }
        }
        if (with_aux && aux_loader) {
            if (UADDR) {
                const uint64_t ab = reinterpret_cast<uint64_t>(apt);
                const uint64_t au = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ab >> 32)) << 32) |
                                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ab);
                // (lanes past the tile's RT rows re-read its first rows: the tile's aux values end where the slab's may)
                uint32_t lo32 = (uint32_t)(RT < 64 ? lane & (RT - 1) : lane) * (SQ8 ? 16u : 4u);
                asm volatile("" : "+v"(lo32));
                if (SQ8) glds16<0>(reinterpret_cast<const char *>(au) + lo32, abuf_i * AUXBUF, aux_lds);
                else glds4(reinterpret_cast<const char *>(au) + lo32, abuf_i * 256, aux_lds);
            }
            else if (SQ8) glds16<0>(apt, abuf_i * AUXBUF, aux_lds);
            else glds4(apt, abuf_i * 256, aux_lds);
        }
    };

    uint32_t tile = pair_map ? ((blockIdx.x >> 4) * 8 + (blockIdx.x & 7u)) : blockIdx.x;
    uint32_t ftile = tile, fbuf = 0;  // frontier: tile and aux buffer of the unit requested next
    make_ptrs(ftile, rp_f, ap_f);
    uint32_t slot_c = 0, abuf = 0;   // ring slot / aux buffer of the unit / tile being consumed
    uint32_t tiles_done = 0;
    auto advance_frontier = [&]() {
        ftile += step;
        make_ptrs(ftile, rp_f, ap_f);
        fbuf = fbuf + 1 == NAUX ? 0 : fbuf + 1;
    };
#pragma unroll
    for (int u = 0; u < D; u++) {
        if (u > 0 && u % KCH == 0) advance_frontier();
        issue(rp_f, ap_f, u % KCH, u, (u % KCH) == 0, fbuf);
    }

    if (SKEW && half == 1) {
        // half 1 runs the same program one barrier later; its share of unit 0 must be in LDS before half 0 starts
        int n_out = (D - 1) * IPW;
#pragma unroll
        for (int j = 1; j < D; j++) n_out += (j % KCH == 0) ? 1 : 0;
        lowp_wait_vmcnt(n_out);
        mf_ring_barrier();
    }
    // dbg bit 3 (diagnosis build): s_memtime stamps split each wave's time into vmcnt wait / barrier / refill request /
    // fragment reads + MFMA issue / epilogue; the sums go to P.tilemin as 8 dwords per wave
    uint32_t ph[5] = {0, 0, 0, 0, 0}, t_prev = 0;
    auto stamp = [&](int i) {
        if (DIAG && (dbg & 8)) {
            const uint32_t now = (uint32_t)__builtin_amdgcn_s_memtime();
            if (i >= 0) ph[i] += now - t_prev;
            t_prev = now;
        }
    };
    stamp(-1);
    // MF_PROBE: this wave's buffered tile minima: [LOWP_PM_TILES][NQW][16 queries] floats behind everything else
    const uint32_t pm_off = mf_lds_offset(lds) + (uint32_t)(lowp_lds_bytes(NWAVES, KSTEPS, RT, NS, STAGE, SKEW)) +
                            (uint32_t)wave * (uint32_t)(LOWP_PM_TILES * NQW * 64) + (uint32_t)m16 * 4u;
    uint32_t pm_n = 0, pm_tile0 = 0;
    auto flush_probe_minima = [&]() {   // lane (kq, m16) writes out the tiles kq, kq + 4, ... of query column m16
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (uint32_t it = (uint32_t)kq; it < pm_n; it += 4) {
#pragma unroll
            for (int nt = 0; nt < NQW; nt++) {
                float v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(pm_off + (uint32_t)((it * NQW + nt) * 64)) : "memory");
                P.tilemin[(size_t)qidx[nt] * P.tilemin_stride + pm_tile0 + it * step] = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores share the VM counter with the ring: one drain per batch
        pm_n = 0;
    };
    for (; tile < P.n_tiles; tile += step) {
        acc_t acc[MT][NQW];
        u32x4_t auxv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NQW; nt++) {
                if constexpr (LK == LP_SQ8) acc[mt][nt] = acc_t{sq_K, sq_K, sq_K, sq_K};   // the epilogue wants D + K, K = 128 sum Y (LowpOps<LP_SQ8>)
                else acc[mt][nt] = acc_t{0, 0, 0, 0};
            }

#pragma unroll
        for (int c = 0; c < KCH; c++) {
            // unit c landed; units c+1 .. c+D-1 (IPW row loads each, +1 aux load where a unit opens a tile)
            // may stay in flight
            if (!SKEW || half == 0) {
                int n_aux = 0;
#pragma unroll
                for (int j = 1; j < D; j++) n_aux += ((c + j) % KCH == 0) ? 1 : 0;
                if (aux_loader) lowp_wait_vmcnt((D - 1) * IPW + n_aux);
                else if (issuer) lowp_wait_vmcnt((D - 1) * IPW);
            }
            stamp(0);
            // SLACK: only the reads of the unit just finished (KSUB*MT of them) may still be queued; everything
            // older -- the slot about to be refilled -- has returned once at most 8 are outstanding
            if (SLACK && KSUB * MT >= 8) asm volatile("s_waitcnt lgkmcnt(8)\n\ts_barrier" ::: "memory");
            else mf_ring_barrier();  // reads of the slot about to be refilled have returned (see mfma_kernels.hpp)
            stamp(1);
            if (!SKEW && MODE == MF_FILTER && c == 0 && (tiles_done & 3u) == 0) {
                // (entries past the queue's capacity go straight to global memory, so a late flush is only slower)
                if (*eq_n >= MF_EQ_CAP / 2) mf_flush_queue<NWAVES * 64>(eq_n, eq, P.counts, P.cand, P.cap);
            }
            auto request_ahead = [&]() {  // unit c+D into the slot unit c-1 occupied
                uint32_t slot_p = slot_c + D;
                if (slot_p >= NS) slot_p -= NS;
                const int kc = (c + D) % KCH;
                if (kc == 0) advance_frontier();
                issue(rp_f, ap_f, kc, slot_p, kc == 0, fbuf);
            };
            // DLATE > 0: the refill is requested after the unit's first DLATE fragments instead of right behind the
            // barrier.  All waves leave the barrier together; with the requests first, every wave queues its DMA
            // pieces on the CU's one address path before its first LDS read and the matrix pipe idles meanwhile.
            if (!SKEW && DLATE == 0) request_ahead();
            stamp(2);
            const char *sbase = lds + slot_c * STAGE;
            // The unit's NFRAG A-fragments are read once each and feed NQW MFMAs.  Left alone hipcc emits
            // ds_read -> s_waitcnt lgkmcnt(0) -> mfma per fragment (measured: 60 % of wave cycles parked, matrix
            // pipe 25 % busy), so the schedule is pinned: PF reads up front, then one read per NQW MFMAs, which
            // keeps PF fragments in flight and lets the compiler count lgkmcnt down instead of draining it.
            constexpr int NFRAG = KSUB * MT;
            auto do_frags = [&](auto f0_tag, auto f1_tag) {
                constexpr int F0 = decltype(f0_tag)::value, F1 = decltype(f1_tag)::value, N = F1 - F0;
                constexpr int PF = N < LOWP_PF ? N : LOWP_PF;  // 8 in flight measured no better (8-wave kernels)
                u32x4_t afr[N];
#pragma unroll
                for (int f = F0; f < F1; f++) {
                    const int j = f / MT, mt = f % MT;
                    const char *rowp = sbase + (mt * 16 + m16) * SEG + (j / 4) * 256;
                    const int p = (4 * (j % 4) + kq) ^ m16;
                    afr[f - F0] = *reinterpret_cast<const u32x4_t *>(rowp + p * 16);
                    if (LK == LP_U8 || LK == LP_U8C) afr[f - F0] ^= 0x80808080u;   // (SQ8 codes are stored re-centred)
                }
#pragma unroll
                for (int f = F0; f < F1; f++) {
                    const int j = f / MT, mt = f % MT;
#pragma unroll
                    for (int nt = 0; nt < NQW; nt++) acc[mt][nt] = Ops::mma(afr[f - F0], qf[nt][c * KSUB + j], acc[mt][nt]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
                for (int f = 0; f < N - PF; f++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, NQW, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, PF * NQW, 0);
            };
            if constexpr (!SKEW && DLATE < 0) {
                // staggered refill: the four waves of a SIMD (w, w+4, w+8, w+12) request at different points of
                // the unit.  Every LDS-DMA piece costs its wave a few hundred cycles of issue (the CU's one
                // address path serialises the pieces); issued by all waves right behind the barrier that is
                // 900 cycles during which no wave feeds the matrix pipe (s_memtime phases)
                constexpr int NG = NWAVES / 4;
                const int grp = wave >> 2;
                static_assert(NG == 2 || NG == 4, "staggered refill: 8 or 16 waves");
                static_assert(NFRAG % NG == 0, "staggered refill: fragments per group");
                constexpr int Q = NFRAG / NG;
                if (grp == 0) request_ahead();
                do_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, Q>{});
                if (grp == 1) request_ahead();
                do_frags(std::integral_constant<int, Q>{}, std::integral_constant<int, 2 * Q>{});
                if constexpr (NG == 4) {
                    if (grp == 2) request_ahead();
                    do_frags(std::integral_constant<int, 2 * Q>{}, std::integral_constant<int, 3 * Q>{});
                    if (grp == 3) request_ahead();
                    do_frags(std::integral_constant<int, 3 * Q>{}, std::integral_constant<int, 4 * Q>{});
                }
            } else if (!(dbg & 2)) {
                if constexpr (!SKEW && DLATE > 0 && DLATE < NFRAG) {
                    do_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, DLATE>{});
                    request_ahead();
                    do_frags(std::integral_constant<int, DLATE>{}, std::integral_constant<int, NFRAG>{});
                } else {
                    do_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, NFRAG>{});
                    if (!SKEW && DLATE > 0) request_ahead();
                }
            } else if (!SKEW && DLATE > 0) request_ahead();
            stamp(3);
            if (c == KCH - 1 && !SQ8) {
                // this tile's aux values (landed with unit 0): plain asm so that hipcc does not tie the read to
                // the LDS-DMA stream and drain vmcnt in front of it
                const uint32_t aoff = aux_lds_off + abuf * 256 + kq * 16;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(auxv[mt]) : "v"(aoff), "n"(mt * 64));
            }
            if (SKEW) {
                // second half-step of the unit: the other half is in its MFMA phase now
                if (half == 1) {  // unit c+1 (half 0 reads it next); this wave has not requested unit c+D yet
                    int n_out = (D - 2) * IPW;
#pragma unroll
                    for (int j = 1; j < D - 1; j++) n_out += ((c + 1 + j) % KCH == 0) ? 1 : 0;
                    lowp_wait_vmcnt(n_out);
                }
                mf_ring_barrier();
                if (c < KCH - 1) request_ahead();   // the tile's last unit requests after the epilogue
            }
            if (!SKEW || c < KCH - 1) slot_c = slot_c + 1 == NS ? 0 : slot_c + 1;
        }

        // ---- epilogue ----
        const uint32_t r0 = tile_row0(tile);
        const uint32_t nvalid = P.n_rows - r0;   // rows of this tile that exist (>= RT except in the last tile)
        bool emitted = false;
        float tmin[NQW];
#pragma unroll
        for (int nt = 0; nt < NQW; nt++) tmin[nt] = INFINITY;
        // aux values were requested by the asm reads above (invisible to hipcc's own lgkmcnt bookkeeping)
        if (SQ8) {
        } else if (MT == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(auxv[0]));
        else if (MT == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(auxv[0]), "+v"(auxv[1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(auxv[0]), "+v"(auxv[1]), "+v"(auxv[2]), "+v"(auxv[3]));
        auto epilogue = [&](auto epi_tag) {
            constexpr int EPI = decltype(epi_tag)::value;
            if (MODE == MF_FILTER) {
                // Survivors are rare (a handful per query in millions of rows): one branch-free pass decides whether
                // ANY lane of the wave has one (same tests as below, minus the row bound, so a superset); the
                // per-value branches of the emitting loop are then skipped for almost every tile.
                bool any = false;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t av = auxv[mt][i];
                        const float rowt = __uint_as_float(av) * fp_rowc;   // (fp kinds: see fpq above)
#pragma unroll
                        for (int nt = 0; nt < NQW; nt++) {
                            if (LK == LP_BF16 || LK == LP_F16) {
                                const float dot = (float)acc[mt][nt][i];
                                if (EPI == LE_FP_L2) any |= !(__builtin_fmaf(-2.0f, dot, rowt) > fpq[nt]);
                                else any |= !((-dot - rowt) > fpq[nt]);
                            } else if (LK == LP_I8 || LK == LP_U8) {
                                const int dot = (int)acc[mt][nt][i];
                                if (EPI == LE_I8_COS) any |= !((float)dot < cosq[nt] * __uint_as_float(av));
                                else if (EPI == LE_I8_L2) any |= (float)((int)av + (int)qaux[nt] - 2 * dot) <= tau[nt];
                                else if (EPI == LE_I8_IP) any |= (float)(1 - dot) <= tau[nt];
                                else any |= (float)(1 - (dot + 128 * (int)av + (int)qaux[nt])) <= tau[nt];
                            } else {
                                const float dot = (float)acc[mt][nt][i];
                                const float ssum = __uint_as_float(av) + __uint_as_float(qaux[nt]);
                                const float a = (EPI == LE_FP_L2) ? (ssum - 2.0f * dot) : (1.0f - dot);
                                any |= !(a - (P.cE * ssum + P.absE) > tau[nt]);
                            }
                        }
                    }
                if (!__any(any)) return;
            }
            if (MODE == MF_PROBE && EPI == LE_I8_COS) {
                // Probe, int8 Cosine: any row's score bounds the tile's minimum from above, so the IEEE divide is spent on one
                // value per lane -- the one with the largest dot / norm, found by cross-multiplication (dot_a n_b > dot_b n_a;
                // exact dots below 2^24) -- instead of on all MT x 4.  A zero-norm row (score NaN) never wins a comparison.
#pragma unroll
                for (int nt = 0; nt < NQW; nt++) {
                    float bd = -3.0e38f, bn = 1.0f;
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const uint32_t lrow = mt * 16 + kq * 4 + i;
                            const float d = (float)(int)acc[mt][nt][i], nx = __uint_as_float(auxv[mt][i]);
                            const bool better = lrow < nvalid && d * bn > bd * nx;
                            bd = better ? d : bd;
                            bn = better ? nx : bn;
                        }
                    const float sc = __fsub_rn(1.0f, __fdiv_rn(bd, __fmul_rn(bn, __uint_as_float(qaux[nt]))));
                    if (bd > -3.0e38f && sc < tmin[nt]) tmin[nt] = sc;
                }
                return;
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t lrow = mt * 16 + kq * 4 + i;
                    const uint32_t av = auxv[mt][i];
#pragma unroll
                    for (int nt = 0; nt < NQW; nt++) {
                        float low, up;
                        if (LK == LP_I8 || LK == LP_U8) {
                            const int dot = (int)acc[mt][nt][i];
                            if (MODE == MF_FILTER && EPI == LE_I8_COS) {
                                // NaN thresholds (zero norms, infinite tau) fall through to the exact test
                                if ((float)dot < cosq[nt] * __uint_as_float(av)) continue;
                            }
                            float sc;
                            if (EPI == LE_I8_L2) sc = (float)((int)av + (int)qaux[nt] - 2 * dot);
                            else if (EPI == LE_I8_IP) sc = (float)(1 - dot);
                            else if (EPI == LE_U8_IP) sc = (float)(1 - (dot + 128 * (int)av + (int)qaux[nt]));
                            else sc = __fsub_rn(1.0f, __fdiv_rn((float)dot, __fmul_rn(__uint_as_float(av), __uint_as_float(qaux[nt]))));
                            low = up = sc;
                        } else {
                            const float dot = (float)acc[mt][nt][i];
                            const float ssum = __uint_as_float(av) + __uint_as_float(qaux[nt]);
                            const float a = (EPI == LE_FP_L2) ? (ssum - 2.0f * dot) : (1.0f - dot);
                            const float E = P.cE * ssum + P.absE;
                            low = a - E;
                            up = a + E;
                        }
                        if (MODE == MF_PROBE) {
                            if (lrow < nvalid && up < tmin[nt]) tmin[nt] = up;
                        } else if (lrow < nvalid && ((LK == LP_I8 || LK == LP_U8) ? (low <= tau[nt]) : !(low > tau[nt]))) {
                            // (fp kinds: a NaN bound -- NaN/Inf next to the row's padded columns -- goes on to the re-rank)
                            const uint32_t row = r0 + lrow;
                            const uint32_t pos = mf_queue_reserve(q_cnt_off);
                            if (SKEW) emitted = true;  // any emission makes the wave drain its queue after the tile
                            if (pos < Q_CAP) {
                                mf_queue_write(q_rec_off + pos * 16, row, (uint32_t)qidx[nt], __float_as_uint(low));
                            } else {
                                uint32_t s = atomicAdd(&P.counts[qidx[nt]], 1u);
                                if (s < P.cap) P.cand[(size_t)qidx[nt] * P.cap + s] = make_uint2(row, __float_as_uint(low));
                                emitted = true;
                            }
                        }
                    }
                }
            }
        };
        // SQ8: bounds on the reference's score from the exact code dot product (LowpOps<LP_SQ8>).  With A = |min y_sum|,
        // B = |delta (s (D + K) + 128 sum e)|:  |score_ref - score| <= g (delta |c'| |e| + delta Wref) + kU (2A + 2B + C), g = 1 (IP) or 2 (L2: the score carries 2 ip), C = 1 (IP) or x_sq + y_sq (L2); kU = 64 ulp
        // covers every fp32 rounding on either side (a dozen at most, each relative to one of those magnitudes).
        // The tile's aux values sit in LDS as four arrays of 64 floats {min, delta, sum_squares, delta |c'|} (k_row_aux_sq8);
        // the accumulators hold D + K.
        auto epilogue_sq8 = [&](auto l2_tag) {
            constexpr bool L2 = decltype(l2_tag)::value;
            constexpr float rG = L2 ? -0.5f : -1.0f;
            const float ysq = sq_c[0];
            const uint32_t arow_off = aux_lds_off + abuf * AUXBUF + (uint32_t)kq * 16u;
            // the four aux arrays' entries for rows mt*16 + kq*4 .. +3
            auto read_aux = [&](int mt, f32x4_t &mn, f32x4_t &dl, f32x4_t &xs, f32x4_t &pp) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(mn) : "v"(arow_off), "n"(mt * 64));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dl) : "v"(arow_off), "n"(256 + mt * 64));
                if (L2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xs) : "v"(arow_off), "n"(512 + mt * 64));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pp) : "v"(arow_off), "n"(768 + mt * 64));
                if (L2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mn), "+v"(dl), "+v"(xs), "+v"(pp));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mn), "+v"(dl), "+v"(pp));
            };
            if (MODE == MF_FILTER) {
                // Screen: survivors are rare, so one branch-free pass decides whether any lane of the wave has a value that may
                // pass; only then does the per-value loop below run.  With g = 2 (L2) or 1 (IP) the screen evaluates
                //     lowS = [x_sq | 1 - shift] + min (-g y_sum) + delta (-g (s f + 128 sum e)) - g delta |c'| |e|
                // in four fused operations on pairs of rows (v_pk_fma_f32: 4 + a conversion + a compare per value, against ~18 for
                // `bounds`) and rejects when lowS > sc_T = tau [- y_sq] + slack.  slack (per query, kernel prologue) holds what
                // the screen leaves out of `bounds`' E with the table-wide maxima in place of the row's values: g max(delta) Wref
                // and 192 ulp of every magnitude involved (64 for E's own term, the rest for the screen's roundings) -- the
                // Cauchy-Schwarz term, which dominates E by orders of magnitude, stays per row.  A NaN anywhere rejects nothing.
                bool any = false;
                const f32x2_t x0 = {1.0f - ysq, 0.0f};   // IP: the score's constant part (ysq holds the shift)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    f32x4_t mn, dl, xs, pp;
                    read_aux(mt, mn, dl, xs, pp);
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const f32x2_t f = {(float)(int)acc[mt][0][2 * h], (float)(int)acc[mt][0][2 * h + 1]};
                        const f32x2_t mn2 = {mn[2 * h], mn[2 * h + 1]}, dl2 = {dl[2 * h], dl[2 * h + 1]}, pp2 = {pp[2 * h], pp[2 * h + 1]};
                        f32x2_t t, s1, s2, lo;
                        // (op_sel picks the dword of a register pair each half of the result reads: the constants are broadcast)
                        asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(t) : "v"(sq_a), "v"(f));
                        if (L2) {
                            const f32x2_t x2 = {xs[2 * h], xs[2 * h + 1]};
                            asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(dl2), "v"(t), "v"(x2));
                        } else {
                            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(s1) : "v"(dl2), "v"(t), "v"(x0));
                        }
                        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(s2) : "v"(mn2), "v"(sq_b), "v"(s1));
                        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(lo) : "v"(pp2), "v"(sq_b), "v"(s2));
                        any |= !(lo[0] > sq_T) | !(lo[1] > sq_T);
                    }
                }
                if (!__any(any)) return;
            }
            if constexpr (MODE == MF_FILTER) {
                // Some lane holds a value the screen did not reject: the same test again value by value, rolled (a rare path --
                // unrolled, its registers would set the allocation of the whole kernel), the value's accumulator picked by
                // compares on the uniform counter.  Every value that passes goes to the exact re-rank; the record's score field
                // is overwritten there (k_exact_pairs).
                static_assert(MT == 4, "SQ8 tiles are 64 rows");
                const float x0s = 1.0f - ysq;
#pragma unroll 1
                for (int v = 0; v < 16; v++) {
                    const int mt = v >> 2, i = v & 3;
                    const acc_t lo4 = mt & 1 ? acc[1][0] : acc[0][0], hi4 = mt & 1 ? acc[3][0] : acc[2][0];
                    const acc_t a4 = mt & 2 ? hi4 : lo4;
                    const int dlo = i & 1 ? a4[1] : a4[0], dhi = i & 1 ? a4[3] : a4[2];
                    const float f = (float)(i & 2 ? dhi : dlo);
                    const uint32_t ao = arow_off + (uint32_t)(mt * 64 + i * 4);
                    float mn, dl, xs = x0s, pp;
                    asm volatile("ds_read_b32 %0, %1" : "=v"(mn) : "v"(ao));
                    asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(dl) : "v"(ao));
                    if (L2) asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(xs) : "v"(ao));
                    asm volatile("ds_read_b32 %0, %1 offset:768" : "=v"(pp) : "v"(ao));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mn), "+v"(dl), "+v"(xs), "+v"(pp));
                    const float t = __fmaf_rn(sq_a[0], f, sq_a[1]);
                    const float s1 = __fmaf_rn(dl, t, xs);
                    const float s2 = __fmaf_rn(mn, sq_b[0], s1);
                    const float lo = __fmaf_rn(pp, sq_b[1], s2);
                    const uint32_t lrow = (uint32_t)(mt * 16 + i) + (uint32_t)kq * 4u;
                    if (lrow < nvalid && !(lo > sq_T)) {   // (a NaN goes on to the exact re-rank)
                        const uint32_t row = r0 + lrow;
                        const uint32_t pos = mf_queue_reserve(q_cnt_off);
                        if (pos < Q_CAP) {
                            mf_queue_write(q_rec_off + pos * 16, row, (uint32_t)qidx[0], __float_as_uint(lo));
                        } else {
                            uint32_t s = atomicAdd(&P.counts[qidx[0]], 1u);
                            if (s < P.cap) P.cand[(size_t)qidx[0] * P.cap + s] = make_uint2(row, __float_as_uint(lo));
                            emitted = true;
                        }
                    }
                }
            } else {
                // Probe: the upper end of every value's bound
                constexpr float kU = 64.0f / 16777216.0f;
                const float Wref = sq_c[1];
                const float qs = rG * sq_a[0], ce = rG * sq_a[1], ysum = rG * sq_b[0], ne = rG * sq_b[1];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    f32x4_t mn4, dl4, xs4 = {0, 0, 0, 0}, pp4;
                    read_aux(mt, mn4, dl4, xs4, pp4);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t lrow = mt * 16 + kq * 4 + i;
                        const float mn = mn4[i], dl = dl4[i], xsq = xs4[i], p = pp4[i];
                        const float f = (float)(int)acc[mt][0][i];
                        const float dq = dl * (qs * f + ce);
                        const float my = mn * ysum;
                        const float ip = my + dq;
                        const float C = L2 ? (xsq + ysq) : 1.0f;
                        const float sc = L2 ? (C - 2.0f * ip) : ((1.0f - ip) - ysq);   // IP: ysq holds the shift (y_mean_ip or 0, exact_kernels.hpp sq8_score)
                        const float E = (L2 ? 2.0f : 1.0f) * (p * ne + dl * Wref) + kU * (2.0f * (fabsf(my) + fabsf(dq)) + C + (L2 ? 0.0f : fabsf(ysq)));   // L2 carries 2 ip
                        const float up = sc + E;
                        if (lrow < nvalid && up < tmin[0]) tmin[0] = up;
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one M-block at a time
                }
            }
        };
        auto epilogue_u8c = [&]() {
            const float nq = qm_r[0];
            const int K = (int)__float_as_uint(qm_r[1]);
            const float tq = tau[0];
            const float omt = 1.0f - tq;
            const float cq = (omt - 1e-5f * (1.0f + fabsf(omt))) * nq;   // screen with a margin, as for int8 Cosine
            const uint32_t arow_off = aux_lds_off + abuf * AUXBUF;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                u32x4_t am[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(am[i]) : "v"(arow_off + (uint32_t)((mt * 16 + kq * 4 + i) * 16)));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(am[0]), "+v"(am[1]), "+v"(am[2]), "+v"(am[3]));
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t lrow = mt * 16 + kq * 4 + i;
                    const float nx = __uint_as_float(am[i][0]);
                    const int dot = (int)acc[mt][0][i] + 128 * (int)am[i][1] + K;
                    if (MODE == MF_FILTER && (float)dot < cq * nx) continue;   // NaN thresholds fall through to the exact test
                    const float sc = __fsub_rn(1.0f, __fdiv_rn((float)dot, __fmul_rn(nx, nq)));
                    if (MODE == MF_PROBE) {
                        if (lrow < nvalid && sc < tmin[0]) tmin[0] = sc;
                    } else if (lrow < nvalid && sc <= tq) {
                        const uint32_t row = r0 + lrow;
                        const uint32_t pos = mf_queue_reserve(q_cnt_off);
                        if (pos < Q_CAP) {
                            mf_queue_write(q_rec_off + pos * 16, row, (uint32_t)qidx[0], __float_as_uint(sc));
                        } else {
                            uint32_t s = atomicAdd(&P.counts[qidx[0]], 1u);
                            if (s < P.cap) P.cand[(size_t)qidx[0] * P.cap + s] = make_uint2(row, __float_as_uint(sc));
                            emitted = true;
                        }
                    }
                }
            }
        };
        if constexpr (U8C) {
            if (!(dbg & 1)) epilogue_u8c();
        } else if constexpr (SQ8) {
            if (!(dbg & 1)) {
                if (P.epi == LE_FP_L2) epilogue_sq8(std::true_type{});
                else epilogue_sq8(std::false_type{});
            }
        } else if (!(dbg & 1)) {
            if (LK == LP_U8) {
                if (P.epi == LE_U8_IP) epilogue(std::integral_constant<int, LE_U8_IP>{});
                else epilogue(std::integral_constant<int, LE_I8_L2>{});
            } else if (LK == LP_I8) {
                if (P.epi == LE_I8_COS) epilogue(std::integral_constant<int, LE_I8_COS>{});
                else if (P.epi == LE_I8_L2) epilogue(std::integral_constant<int, LE_I8_L2>{});
                else epilogue(std::integral_constant<int, LE_I8_IP>{});
            } else {
                if (P.epi == LE_FP_L2) epilogue(std::integral_constant<int, LE_FP_L2>{});
                else epilogue(std::integral_constant<int, LE_FP_IP>{});
            }
        }
        if (MODE == MF_PROBE) {
#pragma unroll
            for (int nt = 0; nt < NQW; nt++) {
                float v = tmin[nt];
                v = fminf(v, __shfl_xor(v, 16));
                v = fminf(v, __shfl_xor(v, 32));
                if (kq == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(pm_off + (uint32_t)((pm_n * NQW + nt) * 64)), "v"(v) : "memory");
            }
            if (pm_n == 0) pm_tile0 = tile;
            if (++pm_n == (uint32_t)LOWP_PM_TILES) flush_probe_minima();
        } else if (SKEW) {
            if (__any(emitted)) drain_wave_queue();
            {   // the request that the last unit postponed (same arithmetic as request_ahead with c = KCH-1)
                uint32_t slot_p = slot_c + D;
                if (slot_p >= NS) slot_p -= NS;
                constexpr int kc = (KCH - 1 + D) % KCH;
                if (kc == 0) advance_frontier();
                issue(rp_f, ap_f, kc, slot_p, kc == 0, fbuf);
                slot_c = slot_c + 1 == NS ? 0 : slot_c + 1;
            }
        } else {
            if (__any(emitted)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        abuf = abuf + 1 == NAUX ? 0 : abuf + 1;
        tiles_done++;
        stamp(4);
    }
    if (DIAG && (dbg & 8) && lane == 0) {
        uint32_t *o = reinterpret_cast<uint32_t *>(P.tilemin) + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NWAVES + wave) * 8;
        for (int i = 0; i < 5; i++) o[i] = ph[i];
        o[5] = tiles_done;
    }
    if (SKEW && half == 0) mf_ring_barrier();  // pairs with half 1's leading barrier
    if (MODE == MF_PROBE && pm_n) flush_probe_minima();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (MODE == MF_FILTER && !SKEW) {
        __builtin_amdgcn_s_barrier();
        mf_flush_queue<NWAVES * 64>(eq_n, eq, P.counts, P.cand, P.cap);
    }
}

// ---- per-row aux values ----
// bf16 / fp16: |x|^2 in double -> float.  kind: 2 = bf16, 3 = fp16 (VSGPU type codes)
static __global__ __launch_bounds__(256) void k_row_norms_h16(const char *rows, uint32_t row_stride, uint32_t dim, uint32_t n,
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
// SQ8: {min, delta, sum_squares, delta |c - 128|_2} of every row, taken out of the (unaligned) metadata behind the codes
// (sum_squares only exists in L2 blobs).  Layout: per group of 64 rows four arrays of 64 floats (1 KiB, fetched by ONE
// LDS-DMA piece of the filter like 16-byte records would be), so that a lane's ds_read_b128 returns the same field of its
// four rows -- operands of v_pk_fma_f32 as they land.  `first` = the in-slab index of rows[0], `out` = the slab's aux base.
// tmax: {max delta, max |min|, max sum_squares} over every row ever stored (float bits; all values >= 0, so unsigned
// order = float order and a NaN wins): the filter's screen takes its table-wide slack terms from them.
static __global__ __launch_bounds__(256) void k_row_aux_sq8(const char *rows, uint32_t row_stride, uint32_t dim, uint32_t n,
                                                     int is_l2, uint32_t first, uint32_t *out, uint32_t *tmax) {
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    const unsigned char *cd = reinterpret_cast<const unsigned char *>(rows + (size_t)row * row_stride);
    const unsigned char *m = cd + dim;
    auto ld = [&](int o) { return (uint32_t)m[o] | ((uint32_t)m[o + 1] << 8) | ((uint32_t)m[o + 2] << 16) | ((uint32_t)m[o + 3] << 24); };
    unsigned long long ss = 0;   // |c - 128|_2, rounded up: the row's factor of the Cauchy-Schwarz term of the filter bound
    for (uint32_t i = 0; i < dim; i++) {
        const int v = (int)(signed char)cd[i];   // (stored as code ^ 0x80 = code - 128: vsgpu.hip sq8_flip_codes)
        ss += (unsigned long long)(v * v);
    }
    const float nc = (float)(sqrt((double)ss) * 1.000001);   // (the factor dwarfs the roundings, the product's below too)
    const uint32_t mn = ld(0), dl = ld(4), xsq = is_l2 ? ld(12) : 0u;
    const uint32_t r = first + row;
    uint32_t *g = out + (size_t)(r >> 6) * 256 + (r & 63u);
    g[0] = mn;
    g[64] = dl;
    g[128] = xsq;
    g[192] = __float_as_uint(fabsf(__uint_as_float(dl)) * nc);
    atomicMax(&tmax[0], dl & 0x7FFFFFFFu);
    atomicMax(&tmax[1], mn & 0x7FFFFFFFu);
    atomicMax(&tmax[2], xsq & 0x7FFFFFFFu);
}
// uint8 Cosine: {stored float norm, sum (x - 128), 0, 0} per row
static __global__ __launch_bounds__(256) void k_row_aux_u8c(const char *rows, uint32_t row_stride, uint32_t dim, uint32_t n, uint4 *out) {
    const int lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const unsigned char *p = reinterpret_cast<const unsigned char *>(rows + (size_t)row * row_stride);
    int s = 0;
    for (uint32_t i = lane; i < dim; i += 64) s += (int)p[i] - 128;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const unsigned char *np = p + dim;
        out[row] = make_uint4((uint32_t)np[0] | ((uint32_t)np[1] << 8) | ((uint32_t)np[2] << 16) | ((uint32_t)np[3] << 24), (uint32_t)s, 0u, 0u);
    }
}
// int8: sum x^2 as int32 (mode 0) or the float norm stored after the elements (mode 1, Cosine rows)
// ext (may be null): {min, max} of the aux values over every row ever stored, signed order of the bit patterns (ints; non-negative
// floats order like their bits); looked at before the atomic, so that only the rows that move an extreme pay for one
static __global__ __launch_bounds__(256) void k_row_aux_i8(const char *rows, uint32_t row_stride, uint32_t dim, uint32_t n,
                                                    int mode, uint32_t *out, int *ext) {
    const int lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const char *p = rows + (size_t)row * row_stride;
    auto note = [&](uint32_t bits) {
        if (!ext) return;
        const int v = (int)bits;
        if (v < __hip_atomic_load(&ext[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&ext[0], v);
        if (v > __hip_atomic_load(&ext[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&ext[1], v);
    };
    if (mode == 1) {
        if (lane == 0) {
            const unsigned char *np = reinterpret_cast<const unsigned char *>(p + dim);
            out[row] = (uint32_t)np[0] | ((uint32_t)np[1] << 8) | ((uint32_t)np[2] << 16) | ((uint32_t)np[3] << 24);
            // the extremes are kept in the signed order of the bit patterns, which is the floats' order for non-negative, non-NaN
            // norms only.  The norm is whatever the blob's tail holds (IP.cpp:264-271 reads it verbatim): a negative or NaN one
            // poisons the pair to {-inf, +inf}, so that x32l_threshold's product is -inf / NaN and EVERY value takes the exact test
            if ((int)out[row] < 0 || out[row] > 0x7F800000u) {
                note(0xFF800000u);
                note(0x7F800000u);
            } else {
                note(out[row]);
            }
        }
        return;
    }
    int s = 0;
    for (uint32_t i = lane; i < dim; i += 64) {
        // modes 2/3: uint8 rows re-centred to x - 128 (sum of squares / plain sum)
        const int v = mode >= 2 ? (int)*reinterpret_cast<const uint8_t *>(p + i) - 128 : (int)*reinterpret_cast<const int8_t *>(p + i);
        s += mode == 3 ? v : v * v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        out[row] = (uint32_t)s;
        note((uint32_t)s);
    }
}

}  // namespace vsg
