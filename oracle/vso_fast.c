/*
 * vso_fast.c -- the CPU baseline leg.  TEST / BENCH INFRASTRUCTURE ONLY (see vso.h).
 *
 * An independent AVX-512 implementation of the same published algorithm as the reference's
 * AVX512F tier (two 16-lane FMA accumulators, residual handled first, halving-tree reduction),
 * written against the Intel intrinsics directly.  It serves two purposes:
 *   1. bench.py's cpu_baseline ("port"): a CPU scan that runs at the speed the reference's
 *      AVX-512 path runs at, timed on the GPU box's host cores;
 *   2. a cross-check of the portable lane emulation in vso.c (tests compare the two bit for bit
 *      when the host has AVX-512F).
 * Every tier of vso.c has an intrinsics twin here: fp32 / fp64 / fp16 two-accumulator AVX-512F kernels, bf16 on
 * AVX512BW + VBMI2 (expand-load head, unpacklo / unpackhi body) and on vdpbf16ps, int8 / uint8 on VNNI (vpdpwssd).
 * The twins follow the published algorithms the reference's kernels implement (cited per function), written from
 * scratch against the Intel intrinsics; tests/test_oracle_kats.py compares them with vso.c bit for bit on random data
 * for every residual class of dim wherever the host CPU has the instructions.
 */
#include "vso.h"
#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int vso_has_avx512(void) {
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx512f") ? 1 : 0;
#else
    return 0;
#endif
}

#if defined(__x86_64__)
#define TGT __attribute__((target("avx512f,avx512bw,avx512vl,fma")))

/* horizontal sum in the gcc-11 _mm512_reduce_add_ps order, spelled out so the result does not
 * depend on which compiler builds this file */
TGT static inline float hsum16(__m512 v) {
    __m256 hi = _mm256_castpd_ps(_mm512_extractf64x4_pd(_mm512_castps_pd(v), 1));
    __m256 lo = _mm512_castps512_ps256(v);
    __m256 t = _mm256_add_ps(hi, lo);
    __m128 h4 = _mm256_extractf128_ps(t, 1), l4 = _mm256_castps256_ps128(t);
    __m128 u = _mm_add_ps(h4, l4);
    __m128 sh = _mm_shuffle_ps(u, u, _MM_SHUFFLE(1, 0, 3, 2));
    __m128 w = _mm_add_ps(u, sh);
    float w0 = _mm_cvtss_f32(w);
    float w1 = _mm_cvtss_f32(_mm_shuffle_ps(w, w, _MM_SHUFFLE(1, 1, 1, 1)));
    return w0 + w1;
}

TGT static float f32_l2_avx512(const float *a, const float *b, size_t d) {
    __m512 acc0 = _mm512_setzero_ps(), acc1 = _mm512_setzero_ps();
    size_t residual = d & 31, head = residual & 15, pos = 0;
    if (head) {
        __mmask16 m = (__mmask16)((1u << head) - 1u);
        __m512 t = _mm512_sub_ps(_mm512_maskz_loadu_ps(m, a), _mm512_maskz_loadu_ps(m, b));
        acc0 = _mm512_mul_ps(t, t);
        pos = head;
    }
    if (residual >= 16) {
        __m512 t = _mm512_sub_ps(_mm512_loadu_ps(a + pos), _mm512_loadu_ps(b + pos));
        acc1 = _mm512_fmadd_ps(t, t, acc1);
        pos += 16;
    }
    for (; pos < d; pos += 32) {
        __m512 t0 = _mm512_sub_ps(_mm512_loadu_ps(a + pos), _mm512_loadu_ps(b + pos));
        acc0 = _mm512_fmadd_ps(t0, t0, acc0);
        __m512 t1 = _mm512_sub_ps(_mm512_loadu_ps(a + pos + 16), _mm512_loadu_ps(b + pos + 16));
        acc1 = _mm512_fmadd_ps(t1, t1, acc1);
    }
    return hsum16(_mm512_add_ps(acc0, acc1));
}

TGT static float f32_ip_avx512(const float *a, const float *b, size_t d) {
    __m512 acc0 = _mm512_setzero_ps(), acc1 = _mm512_setzero_ps();
    size_t residual = d & 31, head = residual & 15, pos = 0;
    if (head) {
        __mmask16 m = (__mmask16)((1u << head) - 1u);
        acc0 = _mm512_mul_ps(_mm512_maskz_loadu_ps(m, a), _mm512_maskz_loadu_ps(m, b));
        pos = head;
    }
    if (residual >= 16) {
        acc1 = _mm512_fmadd_ps(_mm512_loadu_ps(a + pos), _mm512_loadu_ps(b + pos), acc1);
        pos += 16;
    }
    for (; pos < d; pos += 32) {
        acc0 = _mm512_fmadd_ps(_mm512_loadu_ps(a + pos), _mm512_loadu_ps(b + pos), acc0);
        acc1 = _mm512_fmadd_ps(_mm512_loadu_ps(a + pos + 16), _mm512_loadu_ps(b + pos + 16), acc1);
    }
    return 1.0f - hsum16(_mm512_add_ps(acc0, acc1));
}

/* _mm512_reduce_add_pd in gcc 11's order: 8 -> 4 (i, i+4) -> 2 (i, i+2) -> 0 + 1 */
TGT static inline double hsum8d(__m512d v) {
    __m256d t = _mm256_add_pd(_mm512_extractf64x4_pd(v, 1), _mm512_castpd512_pd256(v));
    __m128d u = _mm_add_pd(_mm256_extractf128_pd(t, 1), _mm256_castpd256_pd128(t));
    return _mm_cvtsd_f64(u) + _mm_cvtsd_f64(_mm_unpackhi_pd(u, u));
}

/* fp64, AVX-512F (L2_AVX512F_FP64.h:11-59, IP twin): two 8-lane accumulators, masked head multiplied (not fused) into
 * the first, an optional whole 8-block into the second, then 16 doubles per turn alternating the two */
TGT static double f64_avx512(const double *a, const double *b, size_t d, int l2) {
    __m512d acc0 = _mm512_setzero_pd(), acc1 = _mm512_setzero_pd();
    const size_t residual = d & 15, head = residual & 7;
    size_t pos = 0;
    if (head) {
        const __mmask8 m = (__mmask8)((1u << head) - 1u);
        __m512d x = _mm512_maskz_loadu_pd(m, a), y = _mm512_maskz_loadu_pd(m, b);
        if (l2) x = _mm512_sub_pd(x, y), y = x;
        acc0 = _mm512_mul_pd(x, y);
        pos = head;
    }
    if (residual >= 8) {
        __m512d x = _mm512_loadu_pd(a + pos), y = _mm512_loadu_pd(b + pos);
        if (l2) x = _mm512_sub_pd(x, y), y = x;
        acc1 = _mm512_fmadd_pd(x, y, acc1);
        pos += 8;
    }
    for (; pos < d; pos += 16) {
        __m512d x0 = _mm512_loadu_pd(a + pos), y0 = _mm512_loadu_pd(b + pos);
        __m512d x1 = _mm512_loadu_pd(a + pos + 8), y1 = _mm512_loadu_pd(b + pos + 8);
        if (l2) x0 = _mm512_sub_pd(x0, y0), y0 = x0, x1 = _mm512_sub_pd(x1, y1), y1 = x1;
        acc0 = _mm512_fmadd_pd(x0, y0, acc0);
        acc1 = _mm512_fmadd_pd(x1, y1, acc1);
    }
    const double r = hsum8d(_mm512_add_pd(acc0, acc1));
    return l2 ? r : 1.0 - r;
}

/* fp16, AVX-512F tier (IP_AVX512F_FP16.h:16-68, L2 twin; dim >= 16): vcvtph2ps to 16 floats, the fp32 kernels' shape
 * (masked head multiplied into the first accumulator -- the head is a FULL 16-element load with the lanes past the
 * residual zeroed --, optional 16-block into the second, 32 per turn) */
TGT static float f16_avx512(const uint16_t *a, const uint16_t *b, size_t d, int l2) {
    __m512 acc0 = _mm512_setzero_ps(), acc1 = _mm512_setzero_ps();
    const size_t residual = d & 31, head = residual & 15;
    size_t pos = 0;
    if (head) {
        const __mmask16 m = (__mmask16)((1u << head) - 1u);
        __m512 x = _mm512_maskz_mov_ps(m, _mm512_cvtph_ps(_mm256_loadu_si256((const __m256i *)a)));
        __m512 y = _mm512_maskz_mov_ps(m, _mm512_cvtph_ps(_mm256_loadu_si256((const __m256i *)b)));
        if (l2) x = _mm512_sub_ps(x, y), y = x;
        acc0 = _mm512_mul_ps(x, y);
        pos = head;
    }
    if (residual >= 16) {
        __m512 x = _mm512_cvtph_ps(_mm256_loadu_si256((const __m256i *)(a + pos)));
        __m512 y = _mm512_cvtph_ps(_mm256_loadu_si256((const __m256i *)(b + pos)));
        if (l2) x = _mm512_sub_ps(x, y), y = x;
        acc1 = _mm512_fmadd_ps(x, y, acc1);
        pos += 16;
    }
    for (; pos < d; pos += 32) {
        __m512 x0 = _mm512_cvtph_ps(_mm256_loadu_si256((const __m256i *)(a + pos)));
        __m512 y0 = _mm512_cvtph_ps(_mm256_loadu_si256((const __m256i *)(b + pos)));
        __m512 x1 = _mm512_cvtph_ps(_mm256_loadu_si256((const __m256i *)(a + pos + 16)));
        __m512 y1 = _mm512_cvtph_ps(_mm256_loadu_si256((const __m256i *)(b + pos + 16)));
        if (l2) x0 = _mm512_sub_ps(x0, y0), y0 = x0, x1 = _mm512_sub_ps(x1, y1), y1 = x1;
        acc0 = _mm512_fmadd_ps(x0, y0, acc0);
        acc1 = _mm512_fmadd_ps(x1, y1, acc1);
    }
    const float r = hsum16(_mm512_add_ps(acc0, acc1));
    return l2 ? r : 1.0f - r;
}

/* bf16, AVX512BW + VBMI2 tier (IP_AVX512BW_VBMI2_BF16.h:14-76, L2_AVX512BW_VBMI2_BF16.h:14-78; dim >= 32): ONE 16-lane
 * accumulator; the residual is expand-loaded -- element i of a 16-run lands in the upper half of fp32 lane i --, whole
 * 32-blocks are widened by interleaving zeros below each bf16: unpacklo takes elements {0-3, 8-11, 16-19, 24-27},
 * unpackhi {4-7, 12-15, 20-23, 28-31} */
#define TGT_VBMI2 __attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi2,fma")))
TGT_VBMI2 static float bf16_vbmi2_hw(const uint16_t *a, const uint16_t *b, size_t d, int l2) {
    __m512 acc = _mm512_setzero_ps();
    const size_t residual = d & 31;
    size_t pos = 0;
    const __m512i zero = _mm512_setzero_si512();
    if (residual) {
        size_t left = residual;
        if (left >= 16) {
            __m512 x = _mm512_castsi512_ps(_mm512_maskz_expandloadu_epi16(0xAAAAAAAAu, a));
            __m512 y = _mm512_castsi512_ps(_mm512_maskz_expandloadu_epi16(0xAAAAAAAAu, b));
            if (l2) x = _mm512_sub_ps(x, y), y = x;
            acc = _mm512_fmadd_ps(x, y, acc);
            pos = 16, left -= 16;
        }
        if (left) {
            const __mmask32 m = (__mmask32)(0xAAAAAAAAu & ((1u << (2 * left)) - 1u));
            __m512 x = _mm512_castsi512_ps(_mm512_maskz_expandloadu_epi16(m, a + pos));
            __m512 y = _mm512_castsi512_ps(_mm512_maskz_expandloadu_epi16(m, b + pos));
            if (l2) x = _mm512_sub_ps(x, y), y = x;
            acc = _mm512_fmadd_ps(x, y, acc);
            pos += left;
        }
    }
    for (; pos < d; pos += 32) {
        const __m512i va = _mm512_loadu_si512((const void *)(a + pos)), vb = _mm512_loadu_si512((const void *)(b + pos));
        __m512 x = _mm512_castsi512_ps(_mm512_unpacklo_epi16(zero, va)), y = _mm512_castsi512_ps(_mm512_unpacklo_epi16(zero, vb));
        if (l2) x = _mm512_sub_ps(x, y), y = x;
        acc = _mm512_fmadd_ps(x, y, acc);
        x = _mm512_castsi512_ps(_mm512_unpackhi_epi16(zero, va)), y = _mm512_castsi512_ps(_mm512_unpackhi_epi16(zero, vb));
        if (l2) x = _mm512_sub_ps(x, y), y = x;
        acc = _mm512_fmadd_ps(x, y, acc);
    }
    const float r = hsum16(acc);
    return l2 ? r : 1.0f - r;
}

/* bf16 IP, avx512_bf16 tier (IP_AVX512_BF16_VL_BF16.h:14-47; dim >= 32): vdpbf16ps on a zero-masked head, then whole blocks */
__attribute__((target("avx512f,avx512bw,avx512vl,avx512bf16"))) static float bf16_ip_dpbf16_hw(const uint16_t *a, const uint16_t *b,
                                                                                          size_t d) {
    __m512 acc = _mm512_setzero_ps();
    const size_t residual = d & 31;
    size_t pos = 0;
    if (residual) {
        const __mmask32 m = (__mmask32)((1ull << residual) - 1ull);
        acc = _mm512_dpbf16_ps(acc, (__m512bh)_mm512_maskz_loadu_epi16(m, a), (__m512bh)_mm512_maskz_loadu_epi16(m, b));
        pos = residual;
    }
    for (; pos < d; pos += 32)
        acc = _mm512_dpbf16_ps(acc, (__m512bh)_mm512_loadu_si512((const void *)(a + pos)), (__m512bh)_mm512_loadu_si512((const void *)(b + pos)));
    return 1.0f - hsum16(acc);
}

/* int8 / uint8, VNNI tier (IP_AVX512F_BW_VL_VNNI_INT8.h:11-76, L2 twin :11-65, UINT8 twins; dim >= 32): bytes widened
 * to 16 bits, vpdpwssd into sixteen int32 lanes.  The sums are exact integers, so only the widening (sign / zero) and the
 * epilogue -- 1 - dot as int -> float; Cosine: 1.0f - float(dot) / (norm_a * norm_b) with the norms behind the elements --
 * define the result. */
#define TGT_VNNI __attribute__((target("avx512f,avx512bw,avx512vl,avx512vnni")))
TGT_VNNI static int i8_dot_vnni(const void *a, const void *b, size_t d, int l2, int is_unsigned) {
    __m512i acc = _mm512_setzero_si512();
    const unsigned char *pa = a, *pb = b;
    size_t pos = 0;
    const size_t head = d & 31;
    if (head) {
        const __mmask32 m = (__mmask32)((1ull << head) - 1ull);
        const __m256i xa = _mm256_maskz_loadu_epi8(m, pa), xb = _mm256_maskz_loadu_epi8(m, pb);
        __m512i va = is_unsigned ? _mm512_cvtepu8_epi16(xa) : _mm512_cvtepi8_epi16(xa);
        __m512i vb = is_unsigned ? _mm512_cvtepu8_epi16(xb) : _mm512_cvtepi8_epi16(xb);
        if (l2) va = _mm512_sub_epi16(va, vb), vb = va;
        acc = _mm512_dpwssd_epi32(acc, va, vb);
        pos = head;
    }
    for (; pos < d; pos += 32) {
        const __m256i xa = _mm256_loadu_si256((const __m256i *)(pa + pos)), xb = _mm256_loadu_si256((const __m256i *)(pb + pos));
        __m512i va = is_unsigned ? _mm512_cvtepu8_epi16(xa) : _mm512_cvtepi8_epi16(xa);
        __m512i vb = is_unsigned ? _mm512_cvtepu8_epi16(xb) : _mm512_cvtepi8_epi16(xb);
        if (l2) va = _mm512_sub_epi16(va, vb), vb = va;
        acc = _mm512_dpwssd_epi32(acc, va, vb);
    }
    return _mm512_reduce_add_epi32(acc);
}
TGT_VNNI static float i8_vnni(const void *a, const void *b, size_t d, int metric, int is_unsigned) {
    const int dot = i8_dot_vnni(a, b, d, metric == VSO_L2, is_unsigned);
    if (metric == VSO_L2) return (float)dot;
    if (metric == VSO_IP) return (float)(1 - dot);
    float na, nb;
    memcpy(&na, (const char *)a + d, 4);
    memcpy(&nb, (const char *)b + d, 4);
    const float ip = (float)dot;
    return 1.0f - ip / (na * nb);
}

/* fp16, F16C tier, as the published algorithm runs on real hardware (vcvtph2ps + 256-bit fmadd): four 8-lane
 * accumulators, zero-blended head, whole 8-blocks of the residual into accumulators 1..3, 32 elements per turn of the
 * main loop, lane-wise (0+1)+(2+3), the eight lanes added left to right.  Cross-check for vso.c:f16_f16c. */
__attribute__((target("avx,avx2,fma,f16c"))) static float f16_f16c_hw(const uint16_t *a, const uint16_t *b, size_t d, int l2) {
    __m256 acc[4] = {_mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps(), _mm256_setzero_ps()};
    const size_t residual = d & 31, head = residual & 7;
    size_t pos = 0;
    if (head) {
        float keep[8];
        for (int j = 0; j < 8; j++) keep[j] = (size_t)j < head ? 1.0f : 0.0f;
        const __m256 m = _mm256_cmp_ps(_mm256_loadu_ps(keep), _mm256_setzero_ps(), _CMP_NEQ_OQ);
        __m256 x = _mm256_and_ps(m, _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)a)));
        __m256 y = _mm256_and_ps(m, _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)b)));
        if (l2) {
            __m256 c = _mm256_sub_ps(x, y);
            acc[0] = _mm256_fmadd_ps(c, c, acc[0]);
        } else {
            acc[0] = _mm256_mul_ps(x, y);
        }
        pos = head;
    }
    for (size_t blk = 1; blk <= residual / 8; blk++, pos += 8) {
        __m256 x = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(a + pos)));
        __m256 y = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(b + pos)));
        if (l2) x = _mm256_sub_ps(x, y), y = x;
        acc[blk] = _mm256_fmadd_ps(x, y, acc[blk]);
    }
    while (pos < d)
        for (int k = 0; k < 4; k++, pos += 8) {
            __m256 x = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(a + pos)));
            __m256 y = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i *)(b + pos)));
            if (l2) x = _mm256_sub_ps(x, y), y = x;
            acc[k] = _mm256_fmadd_ps(x, y, acc[k]);
        }
    float t[8];
    _mm256_storeu_ps(t, _mm256_add_ps(_mm256_add_ps(acc[0], acc[1]), _mm256_add_ps(acc[2], acc[3])));
    volatile float r = t[0];  /* (volatile: the eight adds stay scalar and in this order) */
    for (int j = 1; j < 8; j++) r = r + t[j];
    return l2 ? r : 1.0f - r;
}
int vso_has_f16c(void) { return (__builtin_cpu_supports("f16c") && __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2")) ? 1 : 0; }
/* the F16C-tier kernel on this host's vector unit at any dim >= 8; NaN when the host lacks F16C */
double vso_f16c_distance_hw(int metric, size_t dim, const void *a, const void *b) {
    if (!vso_has_f16c() || dim < 8) return NAN;
    return f16_f16c_hw(a, b, dim, metric == VSO_L2);
}

/* one vdpbf16ps on 16 lanes, exposed so the tests can characterise the instruction on hosts
 * that have it (acc, x, y: 16 floats / 32 + 32 bf16) */
__attribute__((target("avx512f,avx512bw,avx512vl,avx512bf16"))) void
vso_probe_dpbf16(float *acc, const uint16_t *x, const uint16_t *y) {
    __m512 s = _mm512_loadu_ps(acc);
    __m512i vx = _mm512_loadu_si512((const void *)x), vy = _mm512_loadu_si512((const void *)y);
    s = _mm512_dpbf16_ps(s, (__m512bh)vx, (__m512bh)vy);
    _mm512_storeu_ps(acc, s);
}
int vso_has_avx512_bf16(void) { return __builtin_cpu_supports("avx512bf16") ? 1 : 0; }
#else
void vso_probe_dpbf16(float *acc, const uint16_t *x, const uint16_t *y) { (void)acc; (void)x; (void)y; }
int vso_has_avx512_bf16(void) { return 0; }
int vso_has_f16c(void) { return 0; }
double vso_f16c_distance_hw(int metric, size_t dim, const void *a, const void *b) { (void)metric; (void)dim; (void)a; (void)b; return NAN; }
#endif

/* host features the twins need */
static int has_all(int want_vbmi2, int want_vnni, int want_bf16) {
#if defined(__x86_64__)
    if (!__builtin_cpu_supports("avx512f") || !__builtin_cpu_supports("avx512bw") || !__builtin_cpu_supports("avx512vl")) return 0;
    if (want_vbmi2 && !__builtin_cpu_supports("avx512vbmi2")) return 0;
    if (want_vnni && !__builtin_cpu_supports("avx512vnni")) return 0;
    if (want_bf16 && !__builtin_cpu_supports("avx512bf16")) return 0;
    return 1;
#else
    return 0;
#endif
}
/* 1 when (type, metric, tier, dim) has an intrinsics twin on this host: the reference's own minimum dims per tier
 * (L2_space.cpp:215-217, 274-276, 329-331, 391-409, 448-450, 504-506; uint8 beyond 33025 elements is scalar) */
int vso_fast_available(int type, int metric, int tier, size_t dim) {
    if (tier == VSO_TIER_SCALAR) return 0;
    switch (type) {
    case VSO_F32: return dim >= 8 && vso_has_avx512();
    case VSO_F64: return dim >= 4 && vso_has_avx512();
    case VSO_F16:
        if (tier == VSO_TIER_AVX512_FP16 && dim >= 32) return 0;   /* half accumulators: no host here executes avx512_fp16 */
        return dim >= 16 && vso_has_avx512();
    case VSO_BF16:
        if (dim < 32) return 0;
        if ((tier == VSO_TIER_AVX512_BF16 || tier == VSO_TIER_AVX512_FP16) && metric != VSO_L2) return has_all(0, 0, 1);
        return has_all(1, 0, 0);
    case VSO_I8: return dim >= 32 && has_all(0, 1, 0);
    case VSO_U8: return dim >= 32 && dim <= 33025 && has_all(0, 1, 0);
    default: return 0;
    }
}
/* single distance through the intrinsics twin of (type, metric, tier); NaN when vso_fast_available says no */
double vso_distance_fast_tier(int type, int metric, int tier, size_t dim, const void *a, const void *b) {
#if defined(__x86_64__)
    if (!vso_fast_available(type, metric, tier, dim)) return NAN;
    const int l2 = metric == VSO_L2;
    switch (type) {
    case VSO_F32: return l2 ? f32_l2_avx512(a, b, dim) : f32_ip_avx512(a, b, dim);
    case VSO_F64: return f64_avx512(a, b, dim, l2);
    case VSO_F16: return f16_avx512(a, b, dim, l2);
    case VSO_BF16:
        if ((tier == VSO_TIER_AVX512_BF16 || tier == VSO_TIER_AVX512_FP16) && !l2) return bf16_ip_dpbf16_hw(a, b, dim);
        return bf16_vbmi2_hw(a, b, dim, l2);
    case VSO_I8: return i8_vnni(a, b, dim, metric, 0);
    case VSO_U8: return i8_vnni(a, b, dim, metric, 1);
    }
#endif
    (void)type; (void)metric; (void)tier; (void)dim; (void)a; (void)b;
    return NAN;
}
double vso_distance_fast(int type, int metric, size_t dim, const void *a, const void *b) {
    return vso_distance_fast_tier(type, metric, VSO_TIER_AVX512, dim, a, b);
}

typedef struct {
    double score;
    size_t label;
} item_t;

static int less_item(const item_t *x, const item_t *y) {
    if (x->score < y->score) return 1;
    if (y->score < x->score) return 0;
    return x->label < y->label;
}

/* sequential top-K of one query, heap kept as a small sorted array (K is tens to hundreds) */
static size_t one_query(int type, int metric, int tier, int fast, size_t dim, const char *rows, size_t n,
                        size_t stride, const void *q, size_t k, size_t *out_l, double *out_s) {
    item_t *h = malloc((k + 1) * sizeof(item_t));
    size_t hs = 0;
    double upper = -INFINITY;
    for (size_t i = 0; i < n; i++) {
        double s;
        if (fast) s = vso_distance_fast_tier(type, metric, tier, dim, rows + i * stride, q);
        else s = vso_distance(type, metric, tier, dim, rows + i * stride, q);
        if (s < upper || hs < k) {
            item_t it = {s, i};
            size_t p = hs++;
            while (p > 0 && less_item(&it, &h[p - 1])) {
                h[p] = h[p - 1];
                p--;
            }
            h[p] = it;
            if (hs > k) hs--; /* drop the largest (score,label) */
            upper = h[hs - 1].score;
        }
    }
    for (size_t i = 0; i < hs; i++) {
        out_l[i] = h[i].label;
        out_s[i] = h[i].score;
    }
    free(h);
    return hs;
}

int vso_flat_topk_batch_fast_tier(int type, int metric, int tier, size_t dim, const void *rows, size_t n,
                                  size_t stride, const void *queries, size_t nq, size_t qstride,
                                  size_t k, int threads, size_t *out_labels, double *out_scores) {
    const int fast = vso_fast_available(type, metric, tier, dim);
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (long qi = 0; qi < (long)nq; qi++) {
        size_t c = one_query(type, metric, tier, fast, dim, rows, n, stride,
                             (const char *)queries + (size_t)qi * qstride, k,
                             out_labels + (size_t)qi * k, out_scores + (size_t)qi * k);
        for (size_t j = c; j < k; j++) {
            out_labels[(size_t)qi * k + j] = (size_t)-1;
            out_scores[(size_t)qi * k + j] = NAN;
        }
    }
    return fast;
}

/* Full-size checker leg: the scores of nq queries against n rows, rows dealt over `threads` OpenMP threads in blocks (each
 * block is read once for all queries).  out[q * n + i] = the same value vso_distance / vso_distance_fast_tier gives for
 * (row i, query q) -- brute_force.h:264-281 calls that function once per stored row; the order of the calls does not enter
 * a score.  Returns 1 if the intrinsics twin ran. */
int vso_scan_batch_fast_tier(int type, int metric, int tier, size_t dim, const void *rows, size_t n, size_t stride,
                             const void *queries, size_t nq, size_t qstride, int threads, double *out) {
    const int fast = vso_fast_available(type, metric, tier, dim);
    const long blk = 128, nblk = (long)((n + blk - 1) / blk);
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
#endif
    for (long b = 0; b < nblk; b++) {
        const size_t i0 = (size_t)b * blk, i1 = i0 + blk < n ? i0 + blk : n;
        for (size_t q = 0; q < nq; q++) {
            const char *qp = (const char *)queries + q * qstride;
            for (size_t i = i0; i < i1; i++) {
                const char *rp = (const char *)rows + i * stride;
                out[q * n + i] = fast ? vso_distance_fast_tier(type, metric, tier, dim, rp, qp)
                                      : vso_distance(type, metric, tier, dim, rp, qp);
            }
        }
    }
    return fast;
}

int vso_flat_topk_batch_fast(int type, int metric, size_t dim, const void *rows, size_t n,
                             size_t stride, const void *queries, size_t nq, size_t qstride,
                             size_t k, int threads, size_t *out_labels, double *out_scores) {
    return vso_flat_topk_batch_fast_tier(type, metric, VSO_TIER_AVX512, dim, rows, n, stride, queries, nq, qstride, k, threads,
                                         out_labels, out_scores);
}
