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
 * Only fp32 (the configuration BASELINE.json's metric is quoted on) has an intrinsics body;
 * other types use the portable code.
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

/* single distance through the intrinsics path (fp32, dim >= 8 only); NaN when unavailable */
double vso_distance_fast(int type, int metric, size_t dim, const void *a, const void *b) {
#if defined(__x86_64__)
    if (type == VSO_F32 && dim >= 8 && vso_has_avx512())
        return metric == VSO_L2 ? f32_l2_avx512(a, b, dim) : f32_ip_avx512(a, b, dim);
#endif
    (void)type; (void)metric; (void)dim; (void)a; (void)b;
    return NAN;
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
static size_t one_query(int type, int metric, int fast, size_t dim, const char *rows, size_t n,
                        size_t stride, const void *q, size_t k, size_t *out_l, double *out_s) {
    item_t *h = malloc((k + 1) * sizeof(item_t));
    size_t hs = 0;
    double upper = -INFINITY;
    for (size_t i = 0; i < n; i++) {
        double s;
#if defined(__x86_64__)
        if (fast)
            s = metric == VSO_L2 ? f32_l2_avx512((const float *)(rows + i * stride), q, dim)
                                 : f32_ip_avx512((const float *)(rows + i * stride), q, dim);
        else
#endif
            s = vso_distance(type, metric, VSO_TIER_AVX512, dim, rows + i * stride, q);
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

int vso_flat_topk_batch_fast(int type, int metric, size_t dim, const void *rows, size_t n,
                             size_t stride, const void *queries, size_t nq, size_t qstride,
                             size_t k, int threads, size_t *out_labels, double *out_scores) {
    int fast = 0;
#if defined(__x86_64__)
    fast = (type == VSO_F32 && dim >= 8 && vso_has_avx512());
#endif
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (long qi = 0; qi < (long)nq; qi++) {
        size_t c = one_query(type, metric, fast, dim, rows, n, stride,
                             (const char *)queries + (size_t)qi * qstride, k,
                             out_labels + (size_t)qi * k, out_scores + (size_t)qi * k);
        for (size_t j = c; j < k; j++) {
            out_labels[(size_t)qi * k + j] = (size_t)-1;
            out_scores[(size_t)qi * k + j] = NAN;
        }
    }
    return fast;
}
