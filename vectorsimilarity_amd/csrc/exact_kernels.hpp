// exact_kernels.hpp -- table-driven exact-distance kernels (gfx950).
//
// One GPU lane plays one accumulator lane of the reference's SIMD kernel (see lane_program.h):
// a group of VL lanes owns R stored rows at a time, walks the lane program step by step with
// element-wide coalesced loads (VL consecutive elements = one contiguous segment per row), keeps
// R x BT accumulators (BT = queries resident in LDS), and finishes with the halving tree.  The
// results are bit-identical to the reference's AVX-512 tier (or scalar tier) by construction.
//
// Used for: small query batches (HBM-bound up to BT ~ 8-16), dense score vectors (batch iterator,
// fallback), range queries, and the exact re-rank of the MFMA filter's survivors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

// never let the compiler fuse or re-associate: the summation order IS the contract
#pragma clang fp contract(off)

namespace vsg {

enum ElemKind { EK_F32 = 0, EK_F64 = 1, EK_BF16 = 2, EK_F16 = 3, EK_I8 = 4, EK_U8 = 5, EK_SQ8 = 6, EK_SQ8H = 7 };
enum OpKind { OP_L2_FMA = 0, OP_IP_FMA = 1, OP_L2_MULADD = 2, OP_IP_MULADD = 3, OP_IP_DPBF16 = 4,
              // AVX512-FP16 tier, fp16 rows: the accumulator is a HALF (carried in a float, every half is one): vsubph + vfmadd...ph
              OP_L2_F16ACC = 5, OP_IP_F16ACC = 6 };
enum ScanMode { MODE_DENSE = 0, MODE_FILTER = 1 };
// how the reduced accumulator becomes a score
enum Epilogue {
    EPI_L2 = 0,         // score = acc
    EPI_ONE_MINUS = 1,  // score = 1 - acc            (IP / fp Cosine: IP.cpp:185-238)
    EPI_INT_L2 = 2,     // score = float(acc)         (L2.cpp:164-174)
    EPI_INT_IP = 3,     // score = float(1 - acc)     (IP.cpp:258-262, 273-277)
    EPI_INT_COS = 4,    // score = 1.0f - float(acc) / (norm_row * norm_q)   (IP.cpp:264-271)
    // SQ8 storage x FP32 query: acc = sum(code_i * y_i);  ip = min * y_sum + delta * acc  (IP.cpp:60-70; fused into
    // fma(min, y_sum, delta * acc) by gcc in the AVX-512 translation unit, see oracle/vso_sq8.c)
    EPI_SQ8_IP = 5,     // score = 1 - ip                                    (IP.cpp:72-80)
    EPI_SQ8_L2 = 6,     // score = (x_sum_sq + y_sum_sq) - 2 ip               (L2.cpp:30-45)
    EPI_ONE_MINUS_H16 = 7   // score = float(half(1) - half(acc))               (IP_AVX512FP16_VL_FP16.h:49-50)
};

template <int EK> struct Elem;
template <> struct Elem<EK_F32> {
    using acc_t = float; using score_t = float;
    static constexpr int VL = 32;
    __device__ static inline float load(const char *p) { return *reinterpret_cast<const float *>(p); }
};
template <> struct Elem<EK_F16> {
    using acc_t = float; using score_t = float;
    static constexpr int VL = 32;
    __device__ static inline float load(const char *p) {
        return (float)(*reinterpret_cast<const _Float16 *>(p));  // exact widening (float16.h:33-52)
    }
};
template <> struct Elem<EK_BF16> {
    using acc_t = float; using score_t = float;
    static constexpr int VL = 16;
    __device__ static inline float load(const char *p) {
        return __uint_as_float((uint32_t)(*reinterpret_cast<const uint16_t *>(p)) << 16);  // bfloat16.h:32-39
    }
};
template <> struct Elem<EK_F64> {
    using acc_t = double; using score_t = double;
    static constexpr int VL = 16;
    __device__ static inline double load(const char *p) { return *reinterpret_cast<const double *>(p); }
};
template <> struct Elem<EK_I8> {
    using acc_t = int; using score_t = float;
    static constexpr int VL = 32;
    __device__ static inline int load(const char *p) { return (int)(*reinterpret_cast<const int8_t *>(p)); }
};
template <> struct Elem<EK_U8> {
    using acc_t = int; using score_t = float;
    static constexpr int VL = 32;
    __device__ static inline int load(const char *p) { return (int)(*reinterpret_cast<const uint8_t *>(p)); }
};
// SQ8 tables keep every code as code ^ 0x80 in HBM, i.e. code - 128 as int8 -- the operand the int8 MFMA of the filter wants
// (vsgpu.hip: sq8_flip_codes on the way in, undone on the way out); code = stored + 128, exact in fp32.
__device__ inline float sq8_code(const char *p) { return (float)((int)(*reinterpret_cast<const int8_t *>(p)) + 128); }
template <> struct Elem<EK_SQ8H> {   // SQ8 rows against fp16 queries: four 16-lane accumulators = 64 virtual lanes
    using acc_t = float; using score_t = float;
    static constexpr int VL = 64;
    __device__ static inline float load(const char *p) { return sq8_code(p); }
};
template <> struct Elem<EK_SQ8> {   // uint8 code widened exactly to float; the query side is fp32
    using acc_t = float; using score_t = float;
    static constexpr int VL = 32;
    __device__ static inline float load(const char *p) { return sq8_code(p); }
};

// vdpbf16ps treats subnormal inputs as zero and flushes subnormal results (IP_AVX512_BF16_VL_BF16.h:14-47; the
// instruction ignores MXCSR: DAZ/FTZ always on)
__device__ inline float ftz_f32(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x7f800000u) ? v : __uint_as_float(u & 0x80000000u);
}
// one accumulation step; explicit rounding intrinsics so no contraction flag can change it
// half-precision steps on values carried as floats: v_sub_f16 / v_fma_f16 / v_add_f16 round to nearest-even and keep subnormals
// (the kernels' default mode for 16- and 64-bit operations), as the AVX512-FP16 instructions do
__device__ inline float h16_fma(float a, float b, float c) { return (float)__builtin_fmaf16((_Float16)a, (_Float16)b, (_Float16)c); }
__device__ inline float h16_sub(float a, float b) { return (float)((_Float16)a - (_Float16)b); }
__device__ inline float h16_add(float a, float b) { return (float)((_Float16)a + (_Float16)b); }
template <int OPK> __device__ inline float acc_step(float x, float q, float acc) {
    if (OPK == OP_L2_F16ACC) { float t = h16_sub(x, q); return h16_fma(t, t, acc); }
    if (OPK == OP_IP_F16ACC) { return h16_fma(x, q, acc); }
    if (OPK == OP_IP_DPBF16) { return ftz_f32(__fmaf_rn(ftz_f32(x), ftz_f32(q), ftz_f32(acc))); }
    if (OPK == OP_L2_FMA) { float t = __fsub_rn(x, q); return __fmaf_rn(t, t, acc); }
    if (OPK == OP_IP_FMA) { return __fmaf_rn(x, q, acc); }
    if (OPK == OP_L2_MULADD) { float t = __fsub_rn(x, q); return __fadd_rn(acc, __fmul_rn(t, t)); }
    return __fadd_rn(acc, __fmul_rn(x, q));
}
template <int OPK> __device__ inline double acc_step(double x, double q, double acc) {
    if (OPK == OP_L2_FMA) { double t = __dsub_rn(x, q); return __fma_rn(t, t, acc); }
    if (OPK == OP_IP_FMA) { return __fma_rn(x, q, acc); }
    if (OPK == OP_L2_MULADD) { double t = __dsub_rn(x, q); return __dadd_rn(acc, __dmul_rn(t, t)); }
    return __dadd_rn(acc, __dmul_rn(x, q));
}
template <int OPK> __device__ inline int acc_step(int x, int q, int acc) {
    static_assert(OPK != OP_IP_DPBF16, "vdpbf16ps order is a bf16 matter");
    if (OPK == OP_L2_FMA || OPK == OP_L2_MULADD) { int t = x - q; return acc + t * t; }
    return acc + x * q;
}

__device__ inline float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ inline double add_rn(double a, double b) { return __dadd_rn(a, b); }
__device__ inline int add_rn(int a, int b) { return a + b; }
__device__ inline long long add_rn(long long a, long long b) { return a + b; }
// what the lanes' accumulators are reduced in: integer rows widen to 64 bits there.  A lane's own int32 sum is safe up to
// ~1 M elements (255^2 * dim / 32), the cross-lane total is not: uint8 rows beyond 33 025 elements are the reference's
// 64-bit scalar kernel (spaces.h:57-66, L2_space.cpp:474-476, IP.cpp:240-262: `long long` total, then float) -- an exact
// integer in any order, so the striding lanes + a 64-bit tree give that number.
template <typename A> struct Reduced { using type = A; };
template <> struct Reduced<int> { using type = long long; };

// Horizontal add of a VL-lane accumulator group; the result is valid in the group's lane 0.
//   kind 0: halving tree, offsets VL/2 .. 1 (sum0 + sum1, then gcc 11's _mm512_reduce_add order)
//   kind 1: the fp16 F16C kernel (dims 8..15): (lane j + lane j+8) + 0 for j < 8, then the eight sums left to right
//           (L2_F16C_FP16.h:81-82, AVX_utils.h:32-37)
template <int VL, typename T> __device__ inline T lane_reduce(T v, int kind) {
    if constexpr (std::is_same<T, float>::value) {
        if (kind == 2) {   // _mm512_reduce_add_ph: the halving tree in half precision
#pragma unroll
            for (int o = VL / 2; o >= 1; o >>= 1) v = h16_add(v, __shfl_down(v, o, VL));
            return v;
        }
    }
    if (kind == 0) {
#pragma unroll
        for (int o = VL / 2; o >= 1; o >>= 1) v = add_rn(v, __shfl_down(v, o, VL));
        return v;
    }
    v = add_rn(add_rn(v, __shfl_down(v, 8, VL)), (T)0);
    T t = __shfl(v, 0, VL);
#pragma unroll
    for (int j = 1; j < 8; j++) t = add_rn(t, __shfl(v, j, VL));
    return t;
}

struct ScanParams {
    // table view: row i lives at slabs[i >> slab_shift] + (i & slab_mask) * row_stride
    const char *const *slabs;
    uint32_t slab_shift;
    uint32_t slab_mask;
    uint32_t row_stride;
    // work: compact row c in [0, n_compact) maps to table row
    //   row_ids ? row_ids[c] : row_begin + (c / tile_rows) * tile_step + (c % tile_rows)
    const uint32_t *row_ids;
    uint32_t row_begin;
    uint32_t row_end;  // exclusive bound on table rows (rows >= row_end are skipped)
    uint32_t n_compact;
    uint32_t tile_step;  // >= tile_rows; == tile_rows for a contiguous scan
    // lane program + queries
    const int32_t *offs;  // [steps][VL]
    int steps;
    int reduce;           // LaneProgram::reduce
    int full_from, full_to;  // every lane is active in steps [full_from, full_to): no predication needed there
    const void *qperm;    // [nq][steps][VL] acc_t, query values already widened & permuted
    int nq;
    // epilogue
    int mode;
    int epilogue;
    uint32_t norm_off;     // byte offset of the row's trailing float norm (EPI_INT_COS) / SQ8 metadata (EPI_SQ8_*)
    const float *qnorm;    // [nq] query norms (EPI_INT_COS); [nq][2] {y_sum, y_sum_squares} (EPI_SQ8_*)
    int sq8_fused;         // EPI_SQ8_*: 1 = the AVX-512 tier's fma(min, y_sum, delta * acc), 0 = the scalar tier's two products
    void *out;             // MODE_DENSE: score_t [nq][out_stride], column = compact row
    size_t out_stride;
    const void *tau;       // MODE_FILTER: score_t [nq]
    uint32_t *counts;      // [nq]
    uint2 *cand;           // [nq][cap] {table row, score bits}
    uint32_t cap;
};

template <typename S> __device__ inline S epilogue_score(long long acc, int epi, float nrow, float nq) {
    if (epi == EPI_INT_L2) return (S)__ll2float_rn(acc);          // float(long long): L2.cpp:164-174
    if (epi == EPI_INT_IP) return (S)__ll2float_rn(1ll - acc);    // IP.cpp:258-262
    float ip = __ll2float_rn(acc);
    return (S)__fsub_rn(1.0f, __fdiv_rn(ip, __fmul_rn(nrow, nq)));
}
template <typename S> __device__ inline S epilogue_score(float acc, int epi, float, float) {
    if (epi == EPI_ONE_MINUS_H16) return (S)h16_sub(1.0f, acc);
    return (epi == EPI_ONE_MINUS) ? (S)__fsub_rn(1.0f, acc) : (S)acc;
}
__device__ inline float load_f32_unaligned(const char *p) {
    const unsigned char *b = reinterpret_cast<const unsigned char *>(p);
    return __uint_as_float((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24));
}
// SQ8 x FP32 score from the reduced code dot product; `meta` = the row's {min, delta, sum, sum_squares} (unaligned).
// y_sum_sq: the query's second metadata slot -- y_sum_squares for L2; for IP the shift DistanceCalculatorWithNorm applies to
// the base score (calculator.h:188-204: base - y_mean_ip; 0 for plain SQ8, and x - 0 is x for every x)
__device__ inline float sq8_score(float qdot, int epi, int fused, const char *meta, float y_sum, float y_sum_sq) {
    const float min_val = load_f32_unaligned(meta), delta = load_f32_unaligned(meta + 4);
    const float dq = __fmul_rn(delta, qdot);
    const float ip = fused ? __fmaf_rn(min_val, y_sum, dq) : __fadd_rn(__fmul_rn(min_val, y_sum), dq);
    if (epi == EPI_SQ8_IP) return __fsub_rn(__fsub_rn(1.0f, ip), y_sum_sq);
    const float x_sq = load_f32_unaligned(meta + 12);
    return __fsub_rn(__fadd_rn(x_sq, y_sum_sq), __fmul_rn(2.0f, ip));
}
template <typename S> __device__ inline S epilogue_score(double acc, int epi, float, float) {
    return (epi == EPI_ONE_MINUS) ? (S)__dsub_rn(1.0, acc) : (S)acc;
}

// R rows per lane group per iteration
template <int EK> struct ScanShape {
    static constexpr int VL = Elem<EK>::VL;
    static constexpr int R = 4;
    static constexpr int GROUPS = 256 / VL;
    static constexpr int TILE_ROWS = GROUPS * R;
};

// GT = true: the lane table and the query image are read from global memory (L1 / L2 hits) instead of an LDS copy -- the
// path of rows whose table + one query image exceed the CU's LDS (any dim is accepted, as the reference does:
// spaces/L2_space.cpp:185-241); one query per pass.
template <int EK, int OPK, int BT, bool GT = false>
__global__ __launch_bounds__(256) void k_exact_scan(ScanParams P) {
    static_assert(!GT || BT == 1, "global-table variant: one query per pass");
    using E = Elem<EK>;
    using acc_t = typename E::acc_t;
    using red_t = typename Reduced<acc_t>::type;
    using score_t = typename E::score_t;
    constexpr int VL = E::VL;
    constexpr int R = ScanShape<EK>::R;

    constexpr int TILE_ROWS = ScanShape<EK>::TILE_ROWS;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int steps = P.steps;
    const int q0 = blockIdx.y * BT;  // first query of this block's tile
    const int nqt = min(BT, P.nq - q0);
    const int32_t *offs_s;
    const acc_t *q_s;
    if constexpr (GT) {
        offs_s = P.offs;
        q_s = reinterpret_cast<const acc_t *>(P.qperm) + (size_t)q0 * steps * VL;
    } else {
        int32_t *offs_l = reinterpret_cast<int32_t *>(smem);
        acc_t *q_l = reinterpret_cast<acc_t *>(smem + (((size_t)P.steps * VL * 4 + 15) & ~(size_t)15));
        for (int i = threadIdx.x; i < steps * VL; i += 256) offs_l[i] = P.offs[i];
        const acc_t *qg = reinterpret_cast<const acc_t *>(P.qperm) + (size_t)q0 * steps * VL;
        for (int i = threadIdx.x; i < BT * steps * VL; i += 256)
            q_l[i] = (i < nqt * steps * VL) ? qg[i] : (acc_t)0;
        __syncthreads();
        offs_s = offs_l;
        q_s = q_l;
    }

    const int lane = threadIdx.x % VL;
    const int grp = threadIdx.x / VL;
    const uint32_t n_tiles = (P.n_compact + TILE_ROWS - 1) / TILE_ROWS;

    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const char *rp[R];
        uint32_t rowid[R];
        uint32_t comp[R];
        bool valid[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            uint32_t c = tile * TILE_ROWS + grp * R + r;
            uint32_t row;
            if (P.row_ids) {
                row = (c < P.n_compact) ? P.row_ids[c] : 0u;
            } else {
                row = P.row_begin + tile * P.tile_step + grp * R + r;
            }
            valid[r] = (c < P.n_compact) && (row < P.row_end);
            if (!valid[r]) row = P.row_begin;  // any mapped row; result discarded
            comp[r] = c;
            rowid[r] = row;
            rp[r] = P.slabs[row >> P.slab_shift] + (size_t)(row & P.slab_mask) * P.row_stride;
        }

        acc_t acc[R][BT];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int b = 0; b < BT; b++) acc[r][b] = (acc_t)0;

        // head: steps where some lanes idle (residual handling) -- predicated
        for (int s = 0; s < P.full_from; s++) {
            const int off = offs_s[s * VL + lane];
            if (off >= 0) {
                acc_t x[R];
#pragma unroll
                for (int r = 0; r < R; r++) x[r] = E::load(rp[r] + off);
#pragma unroll
                for (int b = 0; b < BT; b++) {
                    const acc_t qv = q_s[(b * steps + s) * VL + lane];
#pragma unroll
                    for (int r = 0; r < R; r++) acc[r][b] = acc_step<OPK>(x[r], qv, acc[r][b]);
                }
            }
        }
        // main: all lanes active; CH steps (R*CH row loads) are issued before their FMAs so the loads of a
        // chunk overlap instead of paying one memory round trip per step
        constexpr int CH = 4;
        int s = P.full_from;
        for (; s + CH <= P.full_to; s += CH) {
            acc_t x[CH][R];
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const int off = offs_s[(s + j) * VL + lane];
#pragma unroll
                for (int r = 0; r < R; r++) x[j][r] = E::load(rp[r] + off);
            }
#pragma unroll
            for (int j = 0; j < CH; j++)
#pragma unroll
                for (int b = 0; b < BT; b++) {
                    const acc_t qv = q_s[(b * steps + s + j) * VL + lane];
#pragma unroll
                    for (int r = 0; r < R; r++) acc[r][b] = acc_step<OPK>(x[j][r], qv, acc[r][b]);
                }
        }
        for (; s < steps; s++) {  // leftover full steps and any partial tail
            const int off = offs_s[s * VL + lane];
            if (off >= 0) {
                acc_t x[R];
#pragma unroll
                for (int r = 0; r < R; r++) x[r] = E::load(rp[r] + off);
#pragma unroll
                for (int b = 0; b < BT; b++) {
                    const acc_t qv = q_s[(b * steps + s) * VL + lane];
#pragma unroll
                    for (int r = 0; r < R; r++) acc[r][b] = acc_step<OPK>(x[r], qv, acc[r][b]);
                }
            }
        }

        // halving tree: offsets VL/2 .. 1 (== sum0+sum1, then _mm512_reduce_add order); integer rows in 64 bits
#pragma unroll
        for (int r = 0; r < R; r++) {
            float nrow = 0.f;
            if (lane == 0 && valid[r] && P.epilogue == EPI_INT_COS) nrow = load_f32_unaligned(rp[r] + P.norm_off);
#pragma unroll
            for (int b = 0; b < BT; b++) {
                const red_t tot = lane_reduce<VL>((red_t)acc[r][b], P.reduce);
                if (lane != 0 || !valid[r] || b >= nqt) continue;
                const int q = q0 + b;
                const float nq = (P.epilogue == EPI_INT_COS) ? P.qnorm[q] : 0.f;
                score_t sc;
                if constexpr (EK == EK_SQ8 || EK == EK_SQ8H) {
                    sc = sq8_score(tot, P.epilogue, P.sq8_fused, rp[r] + P.norm_off, P.qnorm[2 * q], P.qnorm[2 * q + 1]);
                } else {
                    sc = epilogue_score<score_t>(tot, P.epilogue, nrow, nq);
                }
                if (P.mode == MODE_DENSE) {
                    reinterpret_cast<score_t *>(P.out)[(size_t)q * P.out_stride + comp[r]] = sc;
                } else {
                    const score_t t = reinterpret_cast<const score_t *>(P.tau)[q];
                    if (sc <= t) {
                        uint32_t slot = atomicAdd(&P.counts[q], 1u);
                        if (slot < P.cap) {
                            uint2 rec;
                            rec.x = rowid[r];
                            rec.y = __float_as_uint((float)sc);
                            P.cand[(size_t)q * P.cap + slot] = rec;
                        }
                    }
                }
            }
        }
    }
}

// ---- probe threshold: tau[q] = k-th smallest of M tile minima of a dense probe score matrix ----
// Any k distinct rows bound the k-th smallest score of the whole table from above, and minima of
// disjoint tiles belong to distinct rows, so tau >= T_q (the exact k-th smallest) always holds.
// One 1024-thread workgroup per query; M <= 8192 (power of two), bitonic sort in LDS.
// klist != nullptr (the streaming filter, mfma_kernels.hpp MF_STREAM): the k smallest group minima seed the query's list
// ([queries][klist_stride] order-preserving integer keys, ascending).
static __global__ __launch_bounds__(1024) void k_probe_threshold(const float *dense, size_t stride,
                                                          uint32_t n0, uint32_t k, uint32_t M,
                                                          float *tau, uint32_t *klist = nullptr, uint32_t klist_stride = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *v = reinterpret_cast<float *>(smem);
    const float *src = dense + (size_t)blockIdx.x * stride;
    const uint32_t ts = (n0 + M - 1) / M;  // rows per tile
    for (uint32_t t = threadIdx.x; t < M; t += 1024) {
        float m = INFINITY;
        const uint32_t lo = t * ts, hi = min(n0, lo + ts);
        for (uint32_t i = lo; i < hi; i++) {
            float s = src[i];
            if (s < m) m = s;  // NaN never wins: a NaN row cannot lower the bound
        }
        v[t] = m;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= M; size <<= 1) {
        for (uint32_t strd = size >> 1; strd > 0; strd >>= 1) {
            for (uint32_t i = threadIdx.x; i < M / 2; i += 1024) {
                uint32_t lo = 2 * i - (i & (strd - 1));
                uint32_t hi = lo + strd;
                bool asc = ((lo & size) == 0);
                float a = v[lo], b = v[hi];
                if ((a > b) == asc) { v[lo] = b; v[hi] = a; }
            }
            __syncthreads();
        }
    }
    // number of non-empty tiles
    const uint32_t tiles = (n0 + ts - 1) / ts;
    if (threadIdx.x == 0) tau[blockIdx.x] = (k >= 1 && k <= tiles) ? v[k - 1] : INFINITY;
    if (klist && threadIdx.x < min(k, klist_stride)) {   // ascending keys; +inf where the probe had fewer than k tiles
        const float f = threadIdx.x < tiles ? v[threadIdx.x] : INFINITY;
        const uint32_t b = __float_as_uint(f);
        klist[(size_t)blockIdx.x * klist_stride + threadIdx.x] = b ^ ((b & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
    }
}

// The same threshold for M > 2048 group minima (k around 100).  // One 1024-thread workgroup per query; M <= 8192 (power of two).  The k-th smallest minimum is found by a bitwise search
// over the order-preserving integer image of the floats (16 counting passes of two bits over <= 8 register-held keys per
// thread, one barrier each) -- the bitonic sort above takes 76 us at M = 8192, this 35-40 us; at M = 1024 the sort is the faster one (15 us).
static __global__ __launch_bounds__(1024) void k_probe_threshold_wide(const float *dense, size_t stride,
                                                          uint32_t n0, uint32_t k, uint32_t M,
                                                          float *tau) {
    __shared__ uint32_t red[2][48];
    const float *src = dense + (size_t)blockIdx.x * stride;
    const uint32_t ts = (n0 + M - 1) / M;  // rows per tile
    uint32_t key[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t t = threadIdx.x + 1024u * (uint32_t)j;
        float m = INFINITY;
        if (t < M) {
            const uint32_t lo = t * ts, hi = min(n0, lo + ts);
            for (uint32_t i = lo; i < hi; i++) {
                float s = src[i];
                if (s < m) m = s;  // NaN never wins: a NaN row cannot lower the bound
            }
        }
        const uint32_t bits = __float_as_uint(m);
        key[j] = (t < M) ? (bits ^ ((bits & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u)) : 0xFFFFFFFFu;   // unsigned order == float order
    }
    const uint32_t tiles = (n0 + ts - 1) / ts;   // non-empty tiles
    if (!(k >= 1 && k <= tiles)) {
        if (threadIdx.x == 0) tau[blockIdx.x] = INFINITY;
        return;
    }
    uint32_t T = 0;
    for (int bit = 30; bit >= 0; bit -= 2) {   // two bits per pass: counts below the three non-zero digits
        uint32_t c1 = 0, c2 = 0, c3 = 0;
        const uint32_t t1 = T | (1u << bit), t2 = T | (2u << bit), t3 = T | (3u << bit);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c1 += (key[j] < t1) ? 1u : 0u;
            c2 += (key[j] < t2) ? 1u : 0u;
            c3 += (key[j] < t3) ? 1u : 0u;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            c1 += __shfl_xor(c1, o);
            c2 += __shfl_xor(c2, o);
            c3 += __shfl_xor(c3, o);
        }
        uint32_t *r = red[(bit >> 1) & 1];   // (alternating buffers: one barrier per pass)
        if ((threadIdx.x & 63) == 0) {
            r[(threadIdx.x >> 6) * 3] = c1;
            r[(threadIdx.x >> 6) * 3 + 1] = c2;
            r[(threadIdx.x >> 6) * 3 + 2] = c3;
        }
        __syncthreads();
        uint32_t s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            s1 += r[w * 3];
            s2 += r[w * 3 + 1];
            s3 += r[w * 3 + 2];
        }
        // fewer than k keys below a trial value: the k-th smallest is at or above it (s1 <= s2 <= s3)
        const uint32_t digit = (s1 < k ? 1u : 0u) + (s2 < k ? 1u : 0u) + (s3 < k ? 1u : 0u);
        T |= digit << bit;
    }
    if (threadIdx.x == 0) {
        const uint32_t bits = (T & 0x80000000u) ? (T ^ 0x80000000u) : ~T;
        tau[blockIdx.x] = __uint_as_float(bits);
    }
}

// (Round 4 tried this threshold by radix selection -- keys in registers, four 8-bit passes over a 256-bin LDS histogram, the
// selection kernel's scheme: parity-clean and no faster per batch on any config within the +-10 us the boxes scatter,
// profiles/r04_thr_radix.txt; removed.)

// ---- synthetic rows (bench / tests): identical to oracle/vso.c:vso_hash32 / vso_synth_f32 ----
__device__ inline uint32_t hash32(uint64_t seed, uint64_t idx) {
    uint64_t x = seed + idx * 0x9E3779B97F4A7C15ull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 32);
}
static __global__ void k_fill_uniform_f32(float *dst, uint64_t first_elem, uint64_t count, uint64_t seed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t strd = (uint64_t)gridDim.x * blockDim.x;
    for (; i < count; i += strd) {
        uint32_t u = hash32(seed, first_elem + i) >> 8;
        dst[i] = __fsub_rn(__fmul_rn((float)u, 1.0f / 8388608.0f), 1.0f);
    }
}

// bf16 / fp16 rows: the fp32 synthetic value rounded to nearest-even (bf16) or converted by the
// hardware RNE cvt (fp16); int8 rows: top byte of the hash.  Twins: vectorsimilarity_amd/synth.py.
static __global__ void k_fill_uniform_h16(uint16_t *dst, uint64_t first_elem, uint64_t count, uint64_t seed, int is_bf16) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t strd = (uint64_t)gridDim.x * blockDim.x;
    for (; i < count; i += strd) {
        uint32_t u = hash32(seed, first_elem + i) >> 8;
        float f = __fsub_rn(__fmul_rn((float)u, 1.0f / 8388608.0f), 1.0f);
        if (is_bf16) {
            uint32_t b = __float_as_uint(f);
            b += 0x7FFFu + ((b >> 16) & 1u);
            dst[i] = (uint16_t)(b >> 16);
        } else {
            _Float16 h = (_Float16)f;
            dst[i] = __builtin_bit_cast(uint16_t, h);
        }
    }
}
// int8 rows, `row_bytes` apart; when with_norm the float norm sqrt(sum x^2) follows the dim bytes
// (compute_norm.h:18-31: uint64 sum, sqrt in double, narrowed to float).  One wave per row.
static __global__ __launch_bounds__(256) void k_fill_rows_i8(char *rows, uint32_t row_bytes, uint32_t dim, uint64_t first_row,
                                                      uint32_t n, uint64_t seed, int with_norm) {
    const int lane = threadIdx.x & 63;
    const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    char *p = rows + (size_t)r * row_bytes;
    unsigned long long ss = 0;
    for (uint32_t i = lane; i < dim; i += 64) {
        const int8_t v = (int8_t)(hash32(seed, (first_row + r) * (uint64_t)dim + i) >> 24);
        p[i] = (char)v;
        ss += (unsigned long long)((int)v * (int)v);
    }
    if (with_norm) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) {
            const float norm = (float)sqrt((double)ss);
            const uint32_t u = __float_as_uint(norm);
            unsigned char *np = reinterpret_cast<unsigned char *>(p + dim);
            np[0] = (unsigned char)u; np[1] = (unsigned char)(u >> 8); np[2] = (unsigned char)(u >> 16); np[3] = (unsigned char)(u >> 24);
        }
    }
}

}  // namespace vsg
