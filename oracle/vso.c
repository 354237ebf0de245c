/*
 * vso.c -- portable C restatement of the reference distance kernels, tier choosers, normalisation
 * and Flat top-K / range scans.  TEST INFRASTRUCTURE ONLY (see vso.h).
 *
 * Build with -O2 -ffp-contract=off (-mfma when the host has it): every FMA of the reference's
 * SIMD kernels is written as an explicit fma()/fmaf(); every place where the reference's scalar
 * translation unit (built without -m flags, spaces/CMakeLists.txt:1-6) does a separate multiply
 * and add is written as two statements that must not be contracted.
 */
#include "vso.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ conversions */

/* types/bfloat16.h:23-30 : round-to-nearest-even on the upper 16 bits, no NaN special case */
uint16_t vso_f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    uint32_t lsb = (u >> 16) & 1u;
    u += 0x7FFFu + lsb;
    return (uint16_t)(u >> 16);
}
/* types/bfloat16.h:32-39 (little endian): bf16 occupies the high half of the fp32 word */
float vso_bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline float bits_f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint32_t f_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
/* types/float16.h:33-52 : exact widening, subnormals via the magic-subtract trick */
float vso_f16_to_f32(uint16_t h) {
    const uint32_t shifted_exp = 0x7c00u << 13;
    int32_t o = ((int32_t)(h & 0x7fffu)) << 13;
    int32_t e = (int32_t)shifted_exp & o;
    o += (int32_t)(127 - 15) << 23;
    int32_t infnan = o + ((int32_t)(128 - 16) << 23);
    float z = bits_f((uint32_t)o + (1u << 23)) - bits_f(113u << 23);
    int32_t zerodenorm = (int32_t)f_bits(z);
    int32_t reg = (e == 0) ? zerodenorm : o;
    int32_t sign = ((int32_t)(h & 0x8000u)) << 16;
    return bits_f((uint32_t)(((e == (int32_t)shifted_exp) ? infnan : reg) | sign));
}
/* types/float16.h:62-117 : truncate 12 low mantissa bits, scale by 2^-112, clamp, add half ulp,
 * shift.  NOT plain IEEE RNE -- restated exactly because Cosine ingest parity depends on it. */
uint16_t vso_f32_to_f16(float f) {
    uint32_t fint = f_bits(f);
    uint32_t sign = fint & 0x80000000u;
    fint ^= sign;
    const uint32_t f32inf = 255u << 23;
    int32_t o = (fint > f32inf) ? 0x7e00 : 0x7c00;
    const uint32_t round_mask = ~0xfffu;
    float fscale = bits_f(fint & round_mask) * bits_f(15u << 23);
    float cap = bits_f((31u << 23) - 0x1000u);
    if (cap < fscale) fscale = cap; /* std::min(fscale, cap) */
    int32_t fint2 = (int32_t)(f_bits(fscale) - round_mask);
    if (fint < f32inf) o = fint2 >> 13;
    return (uint16_t)((uint32_t)o | (sign >> 16));
}

/* array forms (NaN payloads survive: a float never crosses the ctypes boundary by value) */
void vso_f32_to_bf16_n(const float *in, size_t n, uint16_t *out) { for (size_t i = 0; i < n; i++) out[i] = vso_f32_to_bf16(in[i]); }
void vso_f32_to_f16_n(const float *in, size_t n, uint16_t *out) { for (size_t i = 0; i < n; i++) out[i] = vso_f32_to_f16(in[i]); }
void vso_bf16_to_f32_n(const uint16_t *in, size_t n, float *out) { for (size_t i = 0; i < n; i++) out[i] = vso_bf16_to_f32(in[i]); }
void vso_f16_to_f32_n(const uint16_t *in, size_t n, float *out) { for (size_t i = 0; i < n; i++) out[i] = vso_f16_to_f32(in[i]); }

size_t vso_elem_size(int type) {
    switch (type) {
    case VSO_F32: return 4;
    case VSO_F64: return 8;
    case VSO_BF16:
    case VSO_F16: return 2;
    case VSO_I8:
    case VSO_U8: return 1;
    }
    return 0;
}
size_t vso_blob_size(int type, int metric, size_t dim) {
    size_t b = dim * vso_elem_size(type);
    if (metric == VSO_COSINE && (type == VSO_I8 || type == VSO_U8)) b += sizeof(float);
    return b;
}

/* ------------------------------------------------------------------ scalar kernels (L2.cpp / IP.cpp) */

static float f32_l2_scalar(const float *a, const float *b, size_t d) {
    float res = 0;
    for (size_t i = 0; i < d; i++) {
        float t = a[i] - b[i];
        float p = t * t;
        res = res + p;
    }
    return res;
}
static float f32_ip_scalar(const float *a, const float *b, size_t d) {
    float res = 0;
    for (size_t i = 0; i < d; i++) {
        float p = a[i] * b[i];
        res = res + p;
    }
    return 1.0f - res;
}
static double f64_l2_scalar(const double *a, const double *b, size_t d) {
    double res = 0;
    for (size_t i = 0; i < d; i++) {
        double t = a[i] - b[i];
        double p = t * t;
        res = res + p;
    }
    return res;
}
static double f64_ip_scalar(const double *a, const double *b, size_t d) {
    double res = 0;
    for (size_t i = 0; i < d; i++) {
        double p = a[i] * b[i];
        res = res + p;
    }
    return 1.0 - res;
}
typedef float (*widen16_fn)(uint16_t);
static float h16_l2_scalar(const uint16_t *a, const uint16_t *b, size_t d, widen16_fn w) {
    float res = 0;
    for (size_t i = 0; i < d; i++) {
        float t = w(a[i]) - w(b[i]);
        float p = t * t;
        res = res + p;
    }
    return res;
}
static float h16_ip_scalar(const uint16_t *a, const uint16_t *b, size_t d, widen16_fn w) {
    float res = 0;
    for (size_t i = 0; i < d; i++) {
        float p = w(a[i]) * w(b[i]);
        res = res + p;
    }
    return 1.0f - res;
}
/* exact integer sums: L2.cpp:149-174, IP.cpp:247-286.  SIMD tiers produce the same integers
 * (int32 lanes, no overflow below the dims the choosers allow), so one routine serves both. */
static long long i8_dot(const int8_t *a, const int8_t *b, size_t d) {
    long long s = 0;
    for (size_t i = 0; i < d; i++) s += (int)a[i] * (int)b[i];
    return s;
}
static long long u8_dot(const uint8_t *a, const uint8_t *b, size_t d) {
    long long s = 0;
    for (size_t i = 0; i < d; i++) s += (int)a[i] * (int)b[i];
    return s;
}
static long long i8_l2(const int8_t *a, const int8_t *b, size_t d) {
    long long s = 0;
    for (size_t i = 0; i < d; i++) {
        int t = (int)a[i] - (int)b[i];
        s += t * t;
    }
    return s;
}
static long long u8_l2(const uint8_t *a, const uint8_t *b, size_t d) {
    long long s = 0;
    for (size_t i = 0; i < d; i++) {
        int t = (int)a[i] - (int)b[i];
        s += t * t;
    }
    return s;
}
static float load_f32(const void *p) {
    float f;
    memcpy(&f, p, 4);
    return f;
}

/* ------------------------------------------------------------------ AVX-512 tier, lane-exact emulation */

/* _mm512_reduce_add_ps as gcc 11 expands it (avx512fintrin.h:16112-16121):
 * 16 -> 8 (i, i+8) -> 4 (i, i+4) -> {0+2, 1+3} -> sum of the two. */
static float reduce16_f32(const float *v) {
    float t[8], u[4];
    for (int i = 0; i < 8; i++) t[i] = v[i + 8] + v[i];
    for (int i = 0; i < 4; i++) u[i] = t[i + 4] + t[i];
    float w0 = u[0] + u[2];
    float w1 = u[1] + u[3];
    return w0 + w1;
}
/* _mm512_reduce_add_pd: 8 -> 4 (i, i+4) -> 2 (i, i+2) -> 0+1 */
static double reduce8_f64(const double *v) {
    double t[4], u[2];
    for (int i = 0; i < 4; i++) t[i] = v[i + 4] + v[i];
    for (int i = 0; i < 2; i++) u[i] = t[i + 2] + t[i];
    return u[0] + u[1];
}

/* Shared shape of L2_AVX512F_FP32.h:21-59, IP_AVX512F_FP32.h:19-56 and the FP16 AVX512F twins:
 * residual = dim % 32; first residual%16 elements -> sum0 by a plain multiply, then (residual>=16)
 * one 16-block -> sum1 by FMA, then 32 elements per iteration: 16 -> sum0, 16 -> sum1. */
#define GET_F32(p, i) ((p)[i])
#define DEFINE_TWO_ACC_F32(NAME, ELEM_T, WIDEN, IS_L2)                                             \
    static float NAME(const ELEM_T *a, const ELEM_T *b, size_t d) {                                \
        float s0[16], s1[16];                                                                      \
        for (int j = 0; j < 16; j++) s0[j] = s1[j] = 0.0f;                                         \
        size_t residual = d % 32, r16 = residual % 16, pos = 0;                                    \
        if (r16) {                                                                                 \
            for (size_t j = 0; j < r16; j++) {                                                     \
                float x = WIDEN(a[j]), y = WIDEN(b[j]);                                            \
                if (IS_L2) {                                                                       \
                    float t = x - y;                                                               \
                    s0[j] = t * t;                                                                 \
                } else {                                                                           \
                    s0[j] = x * y;                                                                 \
                }                                                                                  \
            }                                                                                      \
            pos = r16;                                                                             \
        }                                                                                          \
        if (residual >= 16) {                                                                      \
            for (int j = 0; j < 16; j++) {                                                         \
                float x = WIDEN(a[pos + j]), y = WIDEN(b[pos + j]);                                \
                if (IS_L2) {                                                                       \
                    float t = x - y;                                                               \
                    s1[j] = fmaf(t, t, s1[j]);                                                     \
                } else {                                                                           \
                    s1[j] = fmaf(x, y, s1[j]);                                                     \
                }                                                                                  \
            }                                                                                      \
            pos += 16;                                                                             \
        }                                                                                          \
        while (pos < d) {                                                                          \
            for (int j = 0; j < 16; j++) {                                                         \
                float x = WIDEN(a[pos + j]), y = WIDEN(b[pos + j]);                                \
                if (IS_L2) {                                                                       \
                    float t = x - y;                                                               \
                    s0[j] = fmaf(t, t, s0[j]);                                                     \
                } else {                                                                           \
                    s0[j] = fmaf(x, y, s0[j]);                                                     \
                }                                                                                  \
            }                                                                                      \
            pos += 16;                                                                             \
            for (int j = 0; j < 16; j++) {                                                         \
                float x = WIDEN(a[pos + j]), y = WIDEN(b[pos + j]);                                \
                if (IS_L2) {                                                                       \
                    float t = x - y;                                                               \
                    s1[j] = fmaf(t, t, s1[j]);                                                     \
                } else {                                                                           \
                    s1[j] = fmaf(x, y, s1[j]);                                                     \
                }                                                                                  \
            }                                                                                      \
            pos += 16;                                                                             \
        }                                                                                          \
        float s[16];                                                                               \
        for (int j = 0; j < 16; j++) s[j] = s0[j] + s1[j];                                         \
        float r = reduce16_f32(s);                                                                 \
        return (IS_L2) ? r : 1.0f - r;                                                             \
    }
#define WIDEN_ID(x) (x)
DEFINE_TWO_ACC_F32(f32_l2_lanes, float, WIDEN_ID, 1)
DEFINE_TWO_ACC_F32(f32_ip_lanes, float, WIDEN_ID, 0)
DEFINE_TWO_ACC_F32(f16_l2_lanes, uint16_t, vso_f16_to_f32, 1)
DEFINE_TWO_ACC_F32(f16_ip_lanes, uint16_t, vso_f16_to_f32, 0)

/* fp16 F16C tier (spaces/L2/L2_F16C_FP16.h:28-83, spaces/IP/IP_F16C_FP16.h:27-81; chooser L2_space.cpp:404-409,
 * IP_space.cpp:664-669: dim >= 8 once the AVX512F tier (dim >= 16) has declined, i.e. dims 8..15 on an AVX-512 host).
 * Four 8-lane fp32 accumulators.  Head: the first residual%8 elements into sum0 through a zero blend (L2: fmadd on a
 * zero accumulator, IP: a plain multiply); then residual/8 whole 8-blocks into sum1, sum2, sum3; then 32 elements per
 * iteration into sum0..sum3.  Lane-wise (sum0+sum1)+(sum2+sum3), then my_mm256_reduce_add_ps (AVX_utils.h:32-37): the
 * eight lanes added left to right. */
static float f16_f16c(const uint16_t *a, const uint16_t *b, size_t d, int l2) {
    float s[4][8];
    for (int k = 0; k < 4; k++)
        for (int j = 0; j < 8; j++) s[k][j] = 0.0f;
    const size_t residual = d % 32, r8 = residual % 8;
    size_t pos = 0;
    if (r8) {
        for (int j = 0; j < 8; j++) {
            float x = (size_t)j < r8 ? vso_f16_to_f32(a[j]) : 0.0f, y = (size_t)j < r8 ? vso_f16_to_f32(b[j]) : 0.0f;
            if (l2) {
                float c = x - y;
                s[0][j] = fmaf(c, c, s[0][j]);
            } else {
                s[0][j] = x * y;
            }
        }
        pos = r8;
    }
    for (size_t blk = 1; blk <= residual / 8; blk++) {
        for (int j = 0; j < 8; j++) {
            float x = vso_f16_to_f32(a[pos + j]), y = vso_f16_to_f32(b[pos + j]);
            if (l2) {
                float c = x - y;
                s[blk][j] = fmaf(c, c, s[blk][j]);
            } else {
                s[blk][j] = fmaf(x, y, s[blk][j]);
            }
        }
        pos += 8;
    }
    while (pos < d) {
        for (int k = 0; k < 4; k++) {
            for (int j = 0; j < 8; j++) {
                float x = vso_f16_to_f32(a[pos + j]), y = vso_f16_to_f32(b[pos + j]);
                if (l2) {
                    float c = x - y;
                    s[k][j] = fmaf(c, c, s[k][j]);
                } else {
                    s[k][j] = fmaf(x, y, s[k][j]);
                }
            }
            pos += 8;
        }
    }
    float t[8];
    for (int j = 0; j < 8; j++) t[j] = (s[0][j] + s[1][j]) + (s[2][j] + s[3][j]);
    float r = t[0] + t[1] + t[2] + t[3] + t[4] + t[5] + t[6] + t[7];
    return l2 ? r : 1.0f - r;
}

/* L2_AVX512F_FP64.h:21-59 and IP twin: residual = dim % 16, 8-lane accumulators */
#define DEFINE_TWO_ACC_F64(NAME, IS_L2)                                                            \
    static double NAME(const double *a, const double *b, size_t d) {                               \
        double s0[8], s1[8];                                                                       \
        for (int j = 0; j < 8; j++) s0[j] = s1[j] = 0.0;                                           \
        size_t residual = d % 16, r8 = residual % 8, pos = 0;                                      \
        if (r8) {                                                                                  \
            for (size_t j = 0; j < r8; j++) {                                                      \
                if (IS_L2) {                                                                       \
                    double t = a[j] - b[j];                                                        \
                    s0[j] = t * t;                                                                 \
                } else {                                                                           \
                    s0[j] = a[j] * b[j];                                                           \
                }                                                                                  \
            }                                                                                      \
            pos = r8;                                                                              \
        }                                                                                          \
        if (residual >= 8) {                                                                       \
            for (int j = 0; j < 8; j++) {                                                          \
                if (IS_L2) {                                                                       \
                    double t = a[pos + j] - b[pos + j];                                            \
                    s1[j] = fma(t, t, s1[j]);                                                      \
                } else {                                                                           \
                    s1[j] = fma(a[pos + j], b[pos + j], s1[j]);                                    \
                }                                                                                  \
            }                                                                                      \
            pos += 8;                                                                              \
        }                                                                                          \
        while (pos < d) {                                                                          \
            for (int j = 0; j < 8; j++) {                                                          \
                if (IS_L2) {                                                                       \
                    double t = a[pos + j] - b[pos + j];                                            \
                    s0[j] = fma(t, t, s0[j]);                                                      \
                } else {                                                                           \
                    s0[j] = fma(a[pos + j], b[pos + j], s0[j]);                                    \
                }                                                                                  \
            }                                                                                      \
            pos += 8;                                                                              \
            for (int j = 0; j < 8; j++) {                                                          \
                if (IS_L2) {                                                                       \
                    double t = a[pos + j] - b[pos + j];                                            \
                    s1[j] = fma(t, t, s1[j]);                                                      \
                } else {                                                                           \
                    s1[j] = fma(a[pos + j], b[pos + j], s1[j]);                                    \
                }                                                                                  \
            }                                                                                      \
            pos += 8;                                                                              \
        }                                                                                          \
        double s[8];                                                                               \
        for (int j = 0; j < 8; j++) s[j] = s0[j] + s1[j];                                          \
        double r = reduce8_f64(s);                                                                 \
        return (IS_L2) ? r : 1.0 - r;                                                              \
    }
DEFINE_TWO_ACC_F64(f64_l2_lanes, 1)
DEFINE_TWO_ACC_F64(f64_ip_lanes, 0)

/* bf16, VBMI2 tier (L2_AVX512BW_VBMI2_BF16.h:14-78, IP_AVX512BW_VBMI2_BF16.h:14-76): ONE 16-lane
 * accumulator.  residual = dim % 32: (residual>=16) elements 0..15 -> lanes 0..15 (expandload with
 * mask 0xAAAAAAAA puts element e into the high half of dword e); then residual%16 elements -> lanes
 * 0..r-1; every op is an FMA into the accumulator.  Full 32-blocks: unpacklo takes, per 128-bit
 * lane L (8 bf16), elements 0..3 -> dwords 4L..4L+3, then unpackhi takes elements 4..7. */
static float bf16_vbmi2(const uint16_t *a, const uint16_t *b, size_t d, int is_l2) {
    float s[16];
    for (int j = 0; j < 16; j++) s[j] = 0.0f;
    size_t residual = d % 32, pos = 0;
    if (residual) {
        if (residual >= 16) {
            for (int j = 0; j < 16; j++) {
                float x = vso_bf16_to_f32(a[j]), y = vso_bf16_to_f32(b[j]);
                if (is_l2) {
                    float t = x - y;
                    s[j] = fmaf(t, t, s[j]);
                } else {
                    s[j] = fmaf(x, y, s[j]);
                }
            }
            pos = 16;
        }
        if (residual != 16) {
            size_t r = residual % 16;
            for (size_t j = 0; j < r; j++) {
                float x = vso_bf16_to_f32(a[pos + j]), y = vso_bf16_to_f32(b[pos + j]);
                if (is_l2) {
                    float t = x - y;
                    s[j] = fmaf(t, t, s[j]);
                } else {
                    s[j] = fmaf(x, y, s[j]);
                }
            }
            pos += r;
        }
    }
    do { /* the reference loop is do{}while: dim >= 32 is guaranteed by the chooser */
        for (int half = 0; half < 2; half++) {
            for (int j = 0; j < 16; j++) {
                int L = j / 4, w = j % 4;
                size_t e = pos + (size_t)(8 * L + 4 * half + w);
                float x = vso_bf16_to_f32(a[e]), y = vso_bf16_to_f32(b[e]);
                if (is_l2) {
                    float t = x - y;
                    s[j] = fmaf(t, t, s[j]);
                } else {
                    s[j] = fmaf(x, y, s[j]);
                }
            }
        }
        pos += 32;
    } while (pos < d);
    float r = reduce16_f32(s);
    return is_l2 ? r : 1.0f - r;
}

/* bf16 IP, avx512_bf16 tier (IP_AVX512_BF16_VL_BF16.h:14-47): vdpbf16ps.  fp32 lane j receives the
 * element pair (2j, 2j+1).  Characterised on an avx512_bf16 host (tests/test_oracle_selfcheck.py
 * re-runs the characterisation when the host has the instruction): per lane
 *     acc = rne(acc + x[2j+1]*y[2j+1]);  acc = rne(acc + x[2j]*y[2j])
 * with bf16 products exact in fp32 and subnormal inputs/outputs flushed to zero. */
static float ftz(float v) {
    uint32_t u = f_bits(v);
    if ((u & 0x7f800000u) == 0) u &= 0x80000000u;
    return bits_f(u);
}
static float dpbf16_lane(float acc, uint16_t x0, uint16_t y0, uint16_t x1, uint16_t y1) {
    float a1 = ftz(vso_bf16_to_f32(x1)), b1 = ftz(vso_bf16_to_f32(y1));
    float a0 = ftz(vso_bf16_to_f32(x0)), b0 = ftz(vso_bf16_to_f32(y0));
    acc = ftz(fmaf(a1, b1, ftz(acc)));
    acc = ftz(fmaf(a0, b0, acc));
    return acc;
}
static float bf16_ip_dpbf16(const uint16_t *a, const uint16_t *b, size_t d) {
    float s[16];
    for (int j = 0; j < 16; j++) s[j] = 0.0f;
    size_t residual = d % 32, pos = 0;
    if (residual) {
        for (int j = 0; j < 16; j++) {
            size_t e0 = 2 * (size_t)j, e1 = e0 + 1;
            uint16_t x0 = e0 < residual ? a[e0] : 0, y0 = e0 < residual ? b[e0] : 0;
            uint16_t x1 = e1 < residual ? a[e1] : 0, y1 = e1 < residual ? b[e1] : 0;
            s[j] = dpbf16_lane(s[j], x0, y0, x1, y1);
        }
        pos = residual;
    }
    do {
        for (int j = 0; j < 16; j++)
            s[j] = dpbf16_lane(s[j], a[pos + 2 * j], b[pos + 2 * j], a[pos + 2 * j + 1],
                               b[pos + 2 * j + 1]);
        pos += 32;
    } while (pos < d);
    return 1.0f - reduce16_f32(s);
}

/* the F16C restatement at any dim >= 8 (the chooser only sends dims 8..15 there on an AVX-512 host) */
double vso_f16c_distance(int metric, size_t dim, const void *a, const void *b) {
    return dim < 8 ? NAN : f16_f16c(a, b, dim, metric == VSO_L2);
}

/* ------------------------------------------------------------------ AVX512-FP16 tier (fp16 rows, fp16 ACCUMULATORS)
 * Builds by gcc >= 12 / clang >= 14 define OPT_AVX512_FP16_VL (cmake/x86_64InstructionFlags.cmake:13,61) and, on hosts with
 * avx512_fp16 && avx512vl, run fp16 rows of dim >= 32 on IP_AVX512FP16_VL_FP16.h:16-51 / L2_AVX512FP16_VL_FP16.h:16-58
 * (choosers IP_space.cpp:649-658, L2_space.cpp:388-397): ONE 32-lane accumulator of HALF precision -- the dim % 32 head by a
 * zero-masked vmulph (L2: zero-masked vsubph, then vmulph), 32 elements per round by vfmadd...ph (L2: vsubph first, rounded to
 * half), _mm512_reduce_add_ph, and for IP `_Float16(1) - res` rounded to half; the float returned is that half widened.
 * _mm512_reduce_add_ph: gcc's avx512fp16intrin.h (_MM512_REDUCE_OP) folds 512 -> 256 -> 128 bits, then the shuffles
 * {4,5,6,7,..}, {2,3,..} and [0] + [1]; clang's (the header is in this image: __builtin_ia32_reduce_fadd_ph512 = a reassociable
 * llvm.vector.reduce.fadd) is expanded by the same halving shuffles: lanes (i, i + 16), (i, i + 8), (i, i + 4), (i, i + 2), 0 + 1.
 * PARITY UNPINNED for this tier: the gcc-11 toolchain of this image cannot emit it, no host here or on the GPU box executes it,
 * and the reference's own test (test_spaces.cpp:1418-1590) holds only "within 1 % of a sequential half-precision sum".
 * Half arithmetic is done exactly: operands as integers in units of 2^-24 (products: 2^-48), one round-to-nearest-even. */
typedef __int128 vso_fx;
static int h_special(uint16_t h) { return (h & 0x7C00) == 0x7C00; }
static vso_fx h_units24(uint16_t h) {
    const int e = (h >> 10) & 31;
    const int64_t m = h & 1023;
    const int64_t v = e ? ((m | 1024) << (e - 1)) : m;   /* (1024 + m) 2^(e - 25) = ((1024 + m) << (e - 1)) 2^-24 */
    return (h & 0x8000) ? -(vso_fx)v : (vso_fx)v;
}
/* v units of 2^-unit_log2 -> nearest half, ties to even; an exact zero takes the sign the caller derived */
static uint16_t h_round(vso_fx v, int unit_log2, int zero_negative) {
    if (v == 0) return zero_negative ? 0x8000 : 0;
    const int neg = v < 0;
    unsigned __int128 m = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    int p = 0;
    for (unsigned __int128 t = m; t >>= 1;) p++;          /* msb index: |value| in [2^(p - unit), 2^(p - unit + 1)) */
    const int e = p - unit_log2;
    const int qe = e < -14 ? -24 : e - 10;               /* the result's quantum: 2^qe (subnormals: 2^-24) */
    const int q = qe + unit_log2;                         /* its bit index in v; >= 0 for units of 2^-24 and 2^-48 */
    if (q > 0) {
        const unsigned __int128 half = (unsigned __int128)1 << (q - 1), rem = m & (((unsigned __int128)1 << q) - 1);
        m >>= q;
        if (rem > half || (rem == half && (m & 1))) m++;
    }
    /* value = m 2^qe, m <= 2048: bits = ((E - 1) << 10) + m with E = qe + 25 (a carry into the exponent adds itself) */
    const unsigned bits = ((unsigned)(qe + 24) << 10) + (unsigned)m;
    return (uint16_t)((neg ? 0x8000u : 0u) | (bits >= 0x7C00u ? 0x7C00u : bits));
}
static uint16_t h_from_special(double r) {   /* a result with an inf / NaN operand, computed in double */
    if (r != r) return 0x7E00;
    if (r == INFINITY) return 0x7C00;
    if (r == -INFINITY) return 0xFC00;
    return vso_f32_to_f16((float)r);   /* (finite: only inf / NaN operands come here, so not reached) */
}
uint16_t vso_h_fma(uint16_t a, uint16_t b, uint16_t c) {
    if (h_special(a) || h_special(b) || h_special(c))
        return h_from_special(fma((double)vso_f16_to_f32(a), (double)vso_f16_to_f32(b), (double)vso_f16_to_f32(c)));
    const vso_fx prod = h_units24(a) * h_units24(b);
    const int prod_neg_zero = prod == 0 && (((a ^ b) & 0x8000) != 0);
    return h_round(prod + h_units24(c) * ((vso_fx)1 << 24), 48, prod_neg_zero && c == 0x8000);
}
uint16_t vso_h_mul(uint16_t a, uint16_t b) {
    if (h_special(a) || h_special(b)) return h_from_special((double)vso_f16_to_f32(a) * (double)vso_f16_to_f32(b));
    return h_round(h_units24(a) * h_units24(b), 48, ((a ^ b) & 0x8000) != 0);
}
uint16_t vso_h_add(uint16_t a, uint16_t b) {
    if (h_special(a) || h_special(b)) return h_from_special((double)vso_f16_to_f32(a) + (double)vso_f16_to_f32(b));
    return h_round(h_units24(a) + h_units24(b), 24, (a & 0x8000) && (b & 0x8000));
}
uint16_t vso_h_sub(uint16_t a, uint16_t b) { return vso_h_add(a, (uint16_t)(b ^ 0x8000)); }

static float f16_fp16acc(const uint16_t *a, const uint16_t *b, size_t d, int l2) {
    uint16_t s[32];
    for (int j = 0; j < 32; j++) s[j] = 0;
    const size_t residual = d % 32;
    size_t pos = 0;
    if (residual) {   /* zero-masked head: the other lanes stay +0 */
        for (size_t j = 0; j < residual; j++) {
            if (l2) {
                const uint16_t t = vso_h_sub(a[j], b[j]);
                s[j] = vso_h_mul(t, t);
            } else {
                s[j] = vso_h_mul(a[j], b[j]);
            }
        }
        pos = residual;
    }
    do {   /* (dim >= 32: at least one whole block) */
        for (int j = 0; j < 32; j++) {
            if (l2) {
                const uint16_t t = vso_h_sub(a[pos + j], b[pos + j]);
                s[j] = vso_h_fma(t, t, s[j]);
            } else {
                s[j] = vso_h_fma(a[pos + j], b[pos + j], s[j]);
            }
        }
        pos += 32;
    } while (pos < d);
    for (int o = 16; o >= 1; o >>= 1)
        for (int j = 0; j < o; j++) s[j] = vso_h_add(s[j], s[j + o]);
    return vso_f16_to_f32(l2 ? s[0] : vso_h_sub(0x3C00, s[0]));
}

/* ------------------------------------------------------------------ tier choosers */

/* Mirrors the x86 branch of L2_space.cpp / IP_space.cpp for a gcc-11 build (no AVX512FP16 tier):
 *   fp32: dim < 8  -> scalar (L2_space.cpp:215-217)     fp64: dim < 4  (:274-276)
 *   bf16: dim < 32 -> scalar (:329-331)                 fp16: dim < 8 -> scalar, 8..15 -> F16C (:404-409), 16+ -> AVX512F (:397)
 *   int8/uint8: exact integers in every tier. */
int vso_uses_scalar(int type, int metric, int tier, size_t dim) {
    (void)metric;
    if (tier == VSO_TIER_SCALAR) return 1;
    switch (type) {
    case VSO_F32: return dim < 8;
    case VSO_F64: return dim < 4;
    case VSO_BF16: return dim < 32;
    case VSO_F16: return dim < 8;  /* 8..15: the F16C tier, 16+: the AVX512F tier */
    default: return 1;
    }
}

double vso_distance(int type, int metric, int tier, size_t dim, const void *a, const void *b) {
    int scalar = vso_uses_scalar(type, metric, tier, dim);
    int l2 = (metric == VSO_L2);
    switch (type) {
    case VSO_F32:
        if (scalar) return l2 ? f32_l2_scalar(a, b, dim) : f32_ip_scalar(a, b, dim);
        return l2 ? f32_l2_lanes(a, b, dim) : f32_ip_lanes(a, b, dim);
    case VSO_F64:
        if (scalar) return l2 ? f64_l2_scalar(a, b, dim) : f64_ip_scalar(a, b, dim);
        return l2 ? f64_l2_lanes(a, b, dim) : f64_ip_lanes(a, b, dim);
    case VSO_F16:
        if (scalar)
            return l2 ? h16_l2_scalar(a, b, dim, vso_f16_to_f32)
                      : h16_ip_scalar(a, b, dim, vso_f16_to_f32);
        if (dim < 16) return f16_f16c(a, b, dim, l2);
        if (tier == VSO_TIER_AVX512_FP16 && dim >= 32) return f16_fp16acc(a, b, dim, l2);
        return l2 ? f16_l2_lanes(a, b, dim) : f16_ip_lanes(a, b, dim);
    case VSO_BF16:
        if (scalar)
            return l2 ? h16_l2_scalar(a, b, dim, vso_bf16_to_f32)
                      : h16_ip_scalar(a, b, dim, vso_bf16_to_f32);
        /* (every x86 CPU with avx512_fp16 has avx512_bf16: the FP16 tier includes the vdpbf16ps kernel) */
        if (!l2 && (tier == VSO_TIER_AVX512_BF16 || tier == VSO_TIER_AVX512_FP16)) return bf16_ip_dpbf16(a, b, dim);
        return bf16_vbmi2(a, b, dim, l2);
    case VSO_I8: {
        if (l2) return (float)i8_l2(a, b, dim);
        long long ip = i8_dot(a, b, dim);
        if (metric == VSO_IP) return (float)(1 - ip);
        float n1 = load_f32((const int8_t *)a + dim), n2 = load_f32((const int8_t *)b + dim);
        return 1.0f - (float)ip / (n1 * n2);
    }
    case VSO_U8: {
        if (l2) return (float)u8_l2(a, b, dim);
        long long ip = u8_dot(a, b, dim);
        if (metric == VSO_IP) return (float)(1 - ip);
        float n1 = load_f32((const uint8_t *)a + dim), n2 = load_f32((const uint8_t *)b + dim);
        return 1.0f - (float)ip / (n1 * n2);
    }
    }
    return NAN;
}

void vso_scan(int type, int metric, int tier, size_t dim, const void *rows, size_t n, size_t stride,
              const void *query, double *out) {
    const char *p = rows;
    for (size_t i = 0; i < n; i++) out[i] = vso_distance(type, metric, tier, dim, p + i * stride, query);
}

/* ------------------------------------------------------------------ normalisation */

void vso_normalize(void *blob, size_t dim, int type) {
    switch (type) {
    case VSO_F32: { /* normalize_naive.h:24-37 : double accumulate, norm cast to float, divide */
        float *v = blob;
        double sum = 0;
        for (size_t i = 0; i < dim; i++) sum += (double)v[i] * (double)v[i];
        float norm = (float)sqrt(sum);
        for (size_t i = 0; i < dim; i++) v[i] = v[i] / norm;
        break;
    }
    case VSO_F64: {
        double *v = blob;
        double sum = 0;
        for (size_t i = 0; i < dim; i++) sum += v[i] * v[i];
        double norm = sqrt(sum);
        for (size_t i = 0; i < dim; i++) v[i] = v[i] / norm;
        break;
    }
    case VSO_BF16:
    case VSO_F16: { /* :40-78 : fp32 accumulate (mul then add), sqrt in double then to float */
        uint16_t *v = blob;
        float *tmp = malloc(dim * sizeof(float) + 4);
        float sum = 0;
        for (size_t i = 0; i < dim; i++) {
            float val = type == VSO_BF16 ? vso_bf16_to_f32(v[i]) : vso_f16_to_f32(v[i]);
            tmp[i] = val;
            float p = val * val;
            sum = sum + p;
        }
        float norm = (float)sqrt((double)sum);
        for (size_t i = 0; i < dim; i++) {
            float q = tmp[i] / norm;
            v[i] = type == VSO_BF16 ? vso_f32_to_bf16(q) : vso_f32_to_f16(q);
        }
        free(tmp);
        break;
    }
    case VSO_I8: { /* :81-88 + compute_norm.h:18-31 : uint64 sum of squares, float norm appended */
        int8_t *v = blob;
        uint64_t sum = 0;
        for (size_t i = 0; i < dim; i++) sum += (uint64_t)((int)v[i] * (int)v[i]);
        float norm = (float)sqrt((double)sum);
        memcpy(v + dim, &norm, 4);
        break;
    }
    case VSO_U8: {
        uint8_t *v = blob;
        uint64_t sum = 0;
        for (size_t i = 0; i < dim; i++) sum += (uint64_t)((int)v[i] * (int)v[i]);
        float norm = (float)sqrt((double)sum);
        memcpy(v + dim, &norm, 4);
        break;
    }
    }
}

/* ------------------------------------------------------------------ Flat scans */

typedef struct {
    double score;
    size_t label;
} heap_item;
/* std::less<pair<score,label>> AS THE REFERENCE'S BUILD HAS IT: gnu++20 (src/VecSim/CMakeLists.txt:15), where pair's `<` is
 * synthesised from operator<=> -- lexicographic on ordinary scores, and on a NaN score the pairs are UNORDERED (neither is
 * less; `first <=> first` is partial_ordering::unordered and the comparison stops there).  C++17's operator< fell through to
 * the labels instead; rounds 1-3 restated that and were wrong on NaN inputs -- found in round 4 by running the reference's
 * own container (oracle/_ref, tests/golden/ref_scalar_random.json section `topk`). */
static int item_less(const heap_item *x, const heap_item *y) {
    if (x->score < y->score) return 1;
    if (y->score < x->score) return 0;
    if (x->score == y->score) return x->label < y->label;
    return 0; /* unordered */
}
/* std::priority_queue<pair<DistType, labelType>> (utils/vecsim_stl.h:66-72) is std::push_heap / std::pop_heap over a
 * vector.  Under a strict weak order any correct heap pops the same items, but a NaN score is unordered (item_less
 * says neither is less), and then WHICH item sits where depends on the heap algorithm itself.  The reference's
 * platform is gcc/libstdc++, so the two routines below follow libstdc++'s published algorithm (bits/stl_heap.h,
 * GCC 11: __push_heap, __adjust_heap, __pop_heap) move for move; tests/test_oracle_kats.py checks them against the
 * real std::priority_queue on inputs with NaNs (tests/helpers/heap_probe.cpp). */
static void heap_push_hole(heap_item *h, size_t hole, size_t top, heap_item value) { /* __push_heap */
    while (hole > top) {
        size_t parent = (hole - 1) / 2;
        if (!item_less(&h[parent], &value)) break;
        h[hole] = h[parent];
        hole = parent;
    }
    h[hole] = value;
}
static void sift_up(heap_item *h, size_t i) { heap_push_hole(h, i, 0, h[i]); } /* push_heap of the appended item */
/* pop_heap + pop_back: h[0 .. n) -> h[0 .. n-1) */
static void heap_pop(heap_item *h, size_t n) {
    if (n < 2) return;
    const heap_item value = h[n - 1]; /* __pop_heap: the last item is re-inserted from the root hole */
    const size_t len = n - 1;
    size_t hole = 0, child = 0;
    while (child < (len - 1) / 2) { /* __adjust_heap: walk the hole down along the larger children ... */
        child = 2 * (child + 1);
        if (item_less(&h[child], &h[child - 1])) child--;
        h[hole] = h[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h[hole] = h[child - 1];
        hole = child - 1;
    }
    heap_push_hole(h, hole, 0, value); /* ... then push the saved item up from there */
}

/* brute_force.h:257-288, NaN scores included: `score < upperBound` is false for a NaN, so a NaN-score row enters only
 * while the heap is not full, and once a NaN is the top nothing else enters. */
size_t vso_topk_replay(const double *scores, const size_t *labels, size_t n, size_t k,
                       size_t *out_labels, double *out_scores) {
    if (k == 0 || n == 0) return 0;
    size_t cap = k < n ? k : n;
    heap_item *h = malloc((cap + 1) * sizeof(heap_item));
    size_t hs = 0;
    double upper = -INFINITY; /* numeric_limits::lowest(); only consulted once the heap is full */
    int upper_init = 0;
    for (size_t i = 0; i < n; i++) {
        double s = scores[i];
        if ((upper_init && s < upper) || hs < k) {
            h[hs].score = s;
            h[hs].label = labels ? labels[i] : i;
            sift_up(h, hs);
            hs++;
            if (hs > k) {
                heap_pop(h, hs);
                hs--;
            }
            upper = h[0].score;
            upper_init = 1;
        }
    }
    size_t cnt = hs;
    for (size_t i = cnt; i-- > 0;) {
        out_labels[i] = h[0].label;
        out_scores[i] = h[0].score;
        heap_pop(h, hs);
        hs--;
    }
    free(h);
    return cnt;
}

/* Multi-value Flat top-K: the same scan with the updatable max-heap of utils/updatable_heap.h:20-113
 * (brute_force_multi.h:108-113): emplace keeps a label's lowest score, pop removes the largest score and,
 * among equal scores, the largest label.  Kept as a plain array (k is small in tests). */
size_t vso_topk_replay_multi(const double *scores, const size_t *labels, size_t n, size_t k,
                             size_t *out_labels, double *out_scores) {
    if (k == 0 || n == 0) return 0;
    heap_item *h = malloc((k + 2) * sizeof(heap_item));
    size_t hs = 0;
    double upper = -INFINITY;
    int upper_init = 0;
    for (size_t i = 0; i < n; i++) {
        double s = scores[i];
        if ((upper_init && s < upper) || hs < k) {
            size_t lab = labels[i], f = hs;
            for (size_t j = 0; j < hs; j++)
                if (h[j].label == lab) { f = j; break; }
            if (f == hs) { h[hs].score = s; h[hs].label = lab; hs++; }
            else if (h[f].score > s) h[f].score = s;
            if (hs > k) {
                size_t m = 0;
                for (size_t j = 1; j < hs; j++)
                    if (h[j].score > h[m].score || (h[j].score == h[m].score && h[j].label > h[m].label)) m = j;
                h[m] = h[--hs];
            }
            upper = h[0].score;
            for (size_t j = 1; j < hs; j++) if (h[j].score > upper) upper = h[j].score;
            upper_init = 1;
        }
    }
    for (size_t i = 0; i < hs; i++)
        for (size_t j = i + 1; j < hs; j++)
            if (h[j].score < h[i].score || (h[j].score == h[i].score && h[j].label < h[i].label)) { heap_item t = h[i]; h[i] = h[j]; h[j] = t; }
    for (size_t i = 0; i < hs; i++) { out_labels[i] = h[i].label; out_scores[i] = h[i].score; }
    size_t c = hs;
    free(h);
    return c;
}

size_t vso_range_replay(const double *scores, const size_t *labels, size_t n, double radius,
                        size_t *out_labels, double *out_scores) {
    size_t c = 0;
    for (size_t i = 0; i < n; i++) {
        if (scores[i] <= radius) {
            out_labels[c] = labels ? labels[i] : i;
            out_scores[c] = scores[i];
            c++;
        }
    }
    return c;
}

size_t vso_flat_topk(int type, int metric, int tier, size_t dim, const void *rows, size_t n,
                     size_t stride, const size_t *labels, const void *query, size_t k,
                     size_t *out_labels, double *out_scores) {
    double *scores = malloc((n ? n : 1) * sizeof(double));
    vso_scan(type, metric, tier, dim, rows, n, stride, query, scores);
    size_t c = vso_topk_replay(scores, labels, n, k, out_labels, out_scores);
    free(scores);
    return c;
}

/* ------------------------------------------------------------------ synthetic data */

uint32_t vso_hash32(uint64_t seed, uint64_t idx) {
    uint64_t x = seed + idx * 0x9E3779B97F4A7C15ull;
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 32);
}
float vso_synth_f32(uint64_t seed, uint64_t idx) {
    uint32_t u = vso_hash32(seed, idx) >> 8; /* 24 bits */
    return (float)u * (1.0f / 8388608.0f) - 1.0f;
}
void vso_synth_rows_f32(uint64_t seed, uint64_t first_elem, size_t count, float *out) {
    for (size_t i = 0; i < count; i++) out[i] = vso_synth_f32(seed, first_elem + i);
}
int8_t vso_synth_i8(uint64_t seed, uint64_t idx) { return (int8_t)(vso_hash32(seed, idx) >> 24); }
