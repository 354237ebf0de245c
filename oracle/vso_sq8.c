/*
 * vso_sq8.c -- CPU ORACLE, SQ8 part: scalar-quantised 8-bit storage (uint8 codes + FP32 metadata), its asymmetric
 * (SQ8 storage x FP32 query) and symmetric (SQ8 x SQ8) distances, and the quantising preprocessor.
 *
 * TEST INFRASTRUCTURE ONLY (see vso.h).  Restated from the reference (paths relative to src/VecSim/):
 *   blob layouts            types/sq8.h:19-62
 *       storage: | codes[dim] u8 | min | delta | sum | sum_squares (L2 only) |     metadata FP32, unaligned
 *       query  : | y[dim] f32    | y_sum | y_sum_squares (L2 only) |
 *   QuantPreprocessor<float, Metric, WithNorm = false>
 *                           spaces/computer/preprocessors.h:259-649 (quantize :270-390, query metadata :398-470)
 *   scalar kernels          spaces/IP/IP.cpp:34-76 (SQ8_FP32), :146-183 (SQ8_SQ8), spaces/L2/L2.cpp:30-45, :185-201
 *   AVX-512 kernels         spaces/IP/IP_AVX512F_BW_VL_VNNI_SQ8_FP32.h:29-120, L2/L2_AVX512F_BW_VL_VNNI_SQ8_FP32.h:30-47,
 *                           IP/IP_AVX512F_BW_VL_VNNI_SQ8_SQ8.h:38-65, L2/L2_AVX512F_BW_VL_VNNI_SQ8_SQ8.h:27-47
 *   tier choice             spaces/L2_space.cpp:41-107, 518-571; spaces/IP_space.cpp:41-176 and the SQ8_SQ8 twins
 *
 * Two things decide the last bits besides the summation order:
 *   * the scalar kernels live in translation units compiled without -m flags (spaces/CMakeLists.txt): baseline x86-64
 *     has no FMA, so  min * y_sum + delta * dot  is two products and one sum, each rounded;
 *   * the AVX-512 kernels are compiled with -mavx512f ... (functions/AVX512F_BW_VL_VNNI.cpp), where gcc's default
 *     -ffp-contract=fast fuses: probed with gcc 11.4 on the same expression shapes,
 *         a*b + c*d                      ->  fma(a, b, c*d)
 *         a*b + c*d + e*f*g - h*i*j      ->  fnma(h*i, j, fma(e*f, g, fma(a, b, c*d)))
 *         x + y - 2*ip                   ->  fnma(2, ip, x + y)   (== (x + y) - 2 ip: 2 ip is exact)
 * FP16 inputs / queries (QuantPreprocessor<float16, ...>, SQ8_FP16_*: IP.cpp:82-144, L2.cpp:47-74,
 * IP/IP_AVX512F_SQ8_FP16.h:29-120, L2/L2_AVX512F_SQ8_FP16.h:18-34, choosers L2_space.cpp:109-180): every fp16 value is
 * widened exactly (types/float16.h:33-52) and all arithmetic is FP32, so the quantiser equals the FP32 one on the widened
 * vector; the AVX-512F kernel keeps FOUR 16-lane accumulators.
 * Mean-centred blobs (QuantPreprocessor<..., WithNorm = true>, preprocessors.h:484-495, 574-640; the calculator on top:
 * spaces/computer/calculator.h:126-232): storage = SQ8 of x - mean (+ x_mean_ip for IP), L2 queries are centred, IP queries
 * stay raw and carry y_mean_ip; distances = the base kernels above plus the calculator's correction terms.  See the
 * *_norm functions at the end.
 */
#include <math.h>
#include <string.h>

#include "vso.h"

static float ldf(const uint8_t *p) {
    float f;
    memcpy(&f, p, 4);
    return f;
}

size_t vso_sq8_storage_size(int metric, size_t dim) { return dim + (metric == VSO_L2 ? 4 : 3) * sizeof(float); }
size_t vso_sq8_query_size(int metric, size_t dim) { return (dim + (metric == VSO_L2 ? 2 : 1)) * sizeof(float); }

/* preprocessors.h:287-299: bounded conversion, then +0.5 and truncate */
static uint8_t to_byte(float scaled) {
    if (!(scaled > 0.0f)) return 0;
    if (scaled >= 255.0f) return 255;
    return (uint8_t)(scaled + 0.5f);
}

/* QuantPreprocessor::quantize (preprocessors.h:270-390) after find_min_max: `t` holds transformed_value() of every
 * element (the input, or input - mean for WithNorm); `tail` = the extra x_mean_ip slot of WithNorm IP blobs, or NULL */
static void quantize_core(const float *t, size_t dim, int metric, float min_val, float max_val, const float *tail,
                          uint8_t *out) {
    const float diff = max_val - min_val;
    const float delta = (diff == 0.0f) ? 1.0f : diff / 255.0f;
    const float inv_delta = 1.0f / delta;
    uint32_t s[4] = {0, 0, 0, 0};
    uint64_t q[4] = {0, 0, 0, 0};
    const size_t d4 = dim & ~(size_t)3;
    size_t i = 0;
    for (; i < d4; i += 4)
        for (int j = 0; j < 4; j++) {
            const uint8_t a = to_byte((t[i + j] - min_val) * inv_delta);
            out[i + j] = a;
            s[j] += a;
            q[j] += (uint64_t)a * a;
        }
    uint32_t q_sum = (s[0] + s[1]) + (s[2] + s[3]);
    uint64_t q_sq = (q[0] + q[1]) + (q[2] + q[3]);
    for (; i < dim; i++) {
        const uint8_t a = to_byte((t[i] - min_val) * inv_delta);
        out[i] = a;
        q_sum += a;
        q_sq += (uint64_t)a * a;
    }
    /* reconstruction sums from the exact byte sums, in double (:369-381); this code is not in an FMA translation unit */
    const double d_min = min_val, d_delta = delta, d_dim = (double)dim;
    const float sum = (float)(d_dim * d_min + d_delta * (double)q_sum);
    float meta[5] = {min_val, delta, sum, 0.0f, 0.0f};
    size_t n = 3;
    if (metric == VSO_L2) {
        const double t0 = d_dim * d_min * d_min;
        const double t1 = 2.0 * d_min * d_delta * (double)q_sum;
        const double t2 = d_delta * d_delta * (double)q_sq;
        meta[n++] = (float)((t0 + t1) + t2);
    }
    if (tail) meta[n++] = *tail;
    memcpy(out + dim, meta, n * sizeof(float));
}

/* DataType = float, WithNorm = false */
void vso_sq8_quantize(const float *x, size_t dim, int metric, uint8_t *out) {
    /* std::minmax_element (:620-622): the first smallest and the LAST largest element */
    float min_val = x[0], max_val = x[0];
    for (size_t i = 1; i < dim; i++) {
        if (x[i] < min_val) min_val = x[i];
        if (!(x[i] < max_val)) max_val = x[i];
    }
    quantize_core(x, dim, metric, min_val, max_val, NULL, out);
}

/* QuantPreprocessor::preprocessQuery + assign_query_metadata (preprocessors.h:398-470, 574-598) */
void vso_sq8_query_blob(const float *y, size_t dim, int metric, float *out) {
    memcpy(out, y, dim * sizeof(float));
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const size_t d4 = dim & ~(size_t)3;
    size_t i = 0;
    for (; i < d4; i += 4)
        for (int j = 0; j < 4; j++) {
            s[j] += y[i + j];
            q[j] += y[i + j] * y[i + j];
        }
    float sum = (s[0] + s[1]) + (s[2] + s[3]);
    float sq = (q[0] + q[1]) + (q[2] + q[3]);
    for (; i < dim; i++) {
        sum += y[i];
        sq += y[i] * y[i];
    }
    out[dim] = sum;
    if (metric == VSO_L2) out[dim + 1] = sq;
}

/* ---- asymmetric: SQ8 storage x FP32 query ---- */
int vso_sq8_fp32_uses_scalar(int tier, size_t dim) { return tier == VSO_TIER_SCALAR || dim < 8; } /* L2_space.cpp:71-75 */

/* sum(code_i * y_i): IP.cpp:34-58 (four chains, separate multiply and add) */
static float qdot_scalar(const uint8_t *c, const float *y, size_t dim) {
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    const size_t d4 = dim & ~(size_t)3;
    size_t i = 0;
    for (; i < d4; i += 4) {
        s0 += (float)c[i] * y[i];
        s1 += (float)c[i + 1] * y[i + 1];
        s2 += (float)c[i + 2] * y[i + 2];
        s3 += (float)c[i + 3] * y[i + 3];
    }
    for (; i < dim; i++) s0 += (float)c[i] * y[i];
    return (s0 + s1) + (s2 + s3);
}
/* IP_AVX512F_BW_VL_VNNI_SQ8_FP32.h:49-104: two 16-lane accumulators; the dim%16 head is a masked MULTIPLY into sum0, a
 * remaining full 16-block of the dim%32 residual goes to sum1, then 32 elements per round (sum0, sum1) with fmadd;
 * sum0 + sum1, then gcc 11's _mm512_reduce_add_ps tree (avx512fintrin.h:16112-16121). */
static float qdot_avx512(const uint8_t *c, const float *y, size_t dim) {
    float acc[32];
    for (int j = 0; j < 32; j++) acc[j] = 0.0f;
    const size_t residual = dim % 32, rh = residual % 16;
    size_t pos = 0;
    for (size_t j = 0; j < rh; j++) acc[j] = (float)c[j] * y[j];
    pos = rh;
    if (residual >= 16) {
        for (size_t j = 0; j < 16; j++) acc[16 + j] = fmaf((float)c[pos + j], y[pos + j], acc[16 + j]);
        pos += 16;
    }
    for (; pos < dim; pos += 32)
        for (size_t j = 0; j < 32; j++) acc[j] = fmaf((float)c[pos + j], y[pos + j], acc[j]);
    float v[16];
    for (int j = 0; j < 16; j++) v[j] = acc[j] + acc[16 + j];
    for (int o = 8; o >= 1; o >>= 1)
        for (int j = 0; j < o; j++) v[j] = v[j] + v[j + o];
    return v[0];
}

/* metric: L2 / IP / Cosine (Cosine == IP on a normalised query: IP.cpp:78-80) */
double vso_sq8_fp32_distance(int metric, int tier, size_t dim, const void *storage, const void *query) {
    const uint8_t *c = (const uint8_t *)storage;
    const float *y = (const float *)query;
    const float min_val = ldf(c + dim), delta = ldf(c + dim + 4);
    const float y_sum = y[dim];
    float ip;
    if (vso_sq8_fp32_uses_scalar(tier, dim)) {
        const float qd = qdot_scalar(c, y, dim);
        const float a = min_val * y_sum, b = delta * qd; /* IP.cpp:70: no FMA in this translation unit */
        ip = a + b;
    } else {
        const float qd = qdot_avx512(c, y, dim);
        ip = fmaf(min_val, y_sum, delta * qd);           /* ...SQ8_FP32.h:103, fused by gcc (header note) */
    }
    if (metric != VSO_L2) return (double)(1.0f - ip);
    const float x_sq = ldf(c + dim + 12), y_sq = y[dim + 1];
    const float t = x_sq + y_sq;
    return (double)(t - 2.0f * ip);                      /* L2.cpp:44 / ...SQ8_FP32.h:46 (2 ip is exact: fused or not alike) */
}

/* ---- symmetric: SQ8 x SQ8 ---- */
/* L2_space.cpp:529-566: scalar beyond the exact 32-bit dim bound, VNNI from dim 64 */
int vso_sq8_sq8_uses_scalar(int tier, size_t dim) { return tier == VSO_TIER_SCALAR || dim < 64 || dim > 33025; }

double vso_sq8_sq8_distance(int metric, int tier, size_t dim, const void *a, const void *b) {
    const uint8_t *p1 = (const uint8_t *)a, *p2 = (const uint8_t *)b;
    const float min1 = ldf(p1 + dim), delta1 = ldf(p1 + dim + 4), sum1 = ldf(p1 + dim + 8);
    const float min2 = ldf(p2 + dim), delta2 = ldf(p2 + dim + 4), sum2 = ldf(p2 + dim + 8);
    float ip;
    if (vso_sq8_sq8_uses_scalar(tier, dim)) {
        float product = 0; /* IP.cpp:152-155: integer product converted, float accumulate */
        for (size_t i = 0; i < dim; i++) product += (float)((int)p1[i] * (int)p2[i]);
        /* IP.cpp:172-173, left to right, no FMA */
        const float t0 = min1 * sum2, t1 = min2 * sum1;
        const float t2 = ((float)dim * min1) * min2;
        const float t3 = (delta1 * delta2) * product;
        ip = ((t0 + t1) - t2) + t3;
    } else {
        int dot = 0; /* UINT8_InnerProductImp: exact int32 */
        for (size_t i = 0; i < dim; i++) dot += (int)p1[i] * (int)p2[i];
        /* ...SQ8_SQ8.h:60-61 under -ffp-contract=fast (header note) */
        const float A = fmaf(min1, sum2, min2 * sum1);
        const float B = fmaf(delta1 * delta2, (float)dot, A);
        ip = fmaf(-((float)dim * min1), min2, B);
    }
    if (metric != VSO_L2) return (double)(1.0f - ip);
    const float sq1 = ldf(p1 + dim + 12), sq2 = ldf(p2 + dim + 12);
    const float t = sq1 + sq2;
    return (double)(t - 2.0f * ip);
}

void vso_sq8_fp32_scan(int metric, int tier, size_t dim, const void *rows, size_t n, size_t stride, const void *query,
                       double *out) {
#pragma omp parallel for schedule(static) if (n * dim > (1u << 22))
    for (size_t i = 0; i < n; i++) out[i] = vso_sq8_fp32_distance(metric, tier, dim, (const char *)rows + i * stride, query);
}

/* ---- FP16 inputs and queries ---- */
size_t vso_sq8_query_size_f16(int metric, size_t dim) { return dim * 2 + (metric == VSO_L2 ? 2 : 1) * sizeof(float); }

/* QuantPreprocessor<float16>::quantize: min / max by FP32 comparison of the widened values (float16.h:54-61),
 * transformed_value = to_fp32 (preprocessors.h:247-252, 611-617): the FP32 quantiser on the widened vector */
void vso_sq8_quantize_f16(const uint16_t *x, size_t dim, int metric, uint8_t *out) {
    float w[dim ? dim : 1];
    for (size_t i = 0; i < dim; i++) w[i] = vso_f16_to_f32(x[i]);
    vso_sq8_quantize(w, dim, metric, out);
}
/* query blob: the fp16 values, then FP32 y_sum (y_sum_squares) of the widened values at an unaligned offset */
void vso_sq8_query_blob_f16(const uint16_t *y, size_t dim, int metric, void *out) {
    memcpy(out, y, dim * 2);
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const size_t d4 = dim & ~(size_t)3;
    size_t i = 0;
    for (; i < d4; i += 4)
        for (int j = 0; j < 4; j++) {
            const float v = vso_f16_to_f32(y[i + j]);
            s[j] += v;
            q[j] += v * v;
        }
    float sum = (s[0] + s[1]) + (s[2] + s[3]);
    float sq = (q[0] + q[1]) + (q[2] + q[3]);
    for (; i < dim; i++) {
        const float v = vso_f16_to_f32(y[i]);
        sum += v;
        sq += v * v;
    }
    float meta[2] = {sum, sq};
    memcpy((char *)out + dim * 2, meta, (metric == VSO_L2 ? 2 : 1) * sizeof(float));
}
int vso_sq8_fp16_uses_scalar(int tier, size_t dim) { return tier == VSO_TIER_SCALAR || dim < 16; } /* L2_space.cpp:121-123 */

double vso_sq8_fp16_distance(int metric, int tier, size_t dim, const void *storage, const void *query) {
    const uint8_t *c = (const uint8_t *)storage;
    const uint16_t *y = (const uint16_t *)query;
    const uint8_t *qm = (const uint8_t *)query + dim * 2;
    const float min_val = ldf(c + dim), delta = ldf(c + dim + 4);
    const float y_sum = ldf(qm);
    float ip;
    if (vso_sq8_fp16_uses_scalar(tier, dim)) {   /* IP.cpp:97-132 */
        float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        const size_t d4 = dim & ~(size_t)3;
        size_t i = 0;
        for (; i < d4; i += 4) {
            s0 += (float)c[i] * vso_f16_to_f32(y[i]);
            s1 += (float)c[i + 1] * vso_f16_to_f32(y[i + 1]);
            s2 += (float)c[i + 2] * vso_f16_to_f32(y[i + 2]);
            s3 += (float)c[i + 3] * vso_f16_to_f32(y[i + 3]);
        }
        for (; i < dim; i++) s0 += (float)c[i] * vso_f16_to_f32(y[i]);
        const float qd = (s0 + s1) + (s2 + s3);
        const float a = min_val * y_sum, b = delta * qd;
        ip = a + b;
    } else {
        /* IP_AVX512F_SQ8_FP16.h:42-101: dim % 16 head = masked multiply into sum0; 64 elements per round over
         * sum0..sum3; up to three 16-chunks of tail into sum0, sum1, sum2; (sum0 + sum1) + (sum2 + sum3); reduce tree */
        float acc[4][16];
        for (int a = 0; a < 4; a++)
            for (int j = 0; j < 16; j++) acc[a][j] = 0.0f;
        const size_t residual = dim % 16;
        size_t pos = 0;
        for (size_t j = 0; j < residual; j++) acc[0][j] = (float)c[j] * vso_f16_to_f32(y[j]);
        pos = residual;
        while (dim - pos >= 64) {
            for (int a = 0; a < 4; a++, pos += 16)
                for (size_t j = 0; j < 16; j++) acc[a][j] = fmaf((float)c[pos + j], vso_f16_to_f32(y[pos + j]), acc[a][j]);
        }
        const size_t remaining = dim - pos;
        for (int a = 0; a < 3; a++)
            if (remaining >= (size_t)(16 * (a + 1))) {
                for (size_t j = 0; j < 16; j++) acc[a][j] = fmaf((float)c[pos + j], vso_f16_to_f32(y[pos + j]), acc[a][j]);
                pos += 16;
            }
        float v[16];
        for (int j = 0; j < 16; j++) v[j] = (acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j]);
        for (int o = 8; o >= 1; o >>= 1)
            for (int j = 0; j < o; j++) v[j] = v[j] + v[j + o];
        ip = fmaf(min_val, y_sum, delta * v[0]);   /* functions/AVX512F.cpp is an FMA translation unit (header note) */
    }
    if (metric != VSO_L2) return (double)(1.0f - ip);
    const float x_sq = ldf(c + dim + 12), y_sq = ldf(qm + 4);
    const float t = x_sq + y_sq;
    return (double)(t - 2.0f * ip);
}
void vso_sq8_fp16_scan(int metric, int tier, size_t dim, const void *rows, size_t n, size_t stride, const void *query,
                       double *out) {
#pragma omp parallel for schedule(static) if (n * dim > (1u << 22))
    for (size_t i = 0; i < n; i++) out[i] = vso_sq8_fp16_distance(metric, tier, dim, (const char *)rows + i * stride, query);
}

/* ------------------------------------------------------------------ mean-centred blobs (WithNorm = true; L2 and IP only)
 * types/sq8.h:38-58: IP blobs grow by one FP32 slot -- storage | codes | min | delta | sum | x_mean_ip |, query
 * | y | y_sum | y_mean_ip | -- and L2 blobs keep their layout. */
size_t vso_sq8_storage_size_norm(int metric, size_t dim) { return dim + 4 * sizeof(float); }
size_t vso_sq8_query_size_norm(int metric, size_t dim) { (void)metric; return (dim + 2) * sizeof(float); }
size_t vso_sq8_query_size_norm_f16(int metric, size_t dim) { (void)metric; return dim * 2 + 2 * sizeof(float); }

/* find_min_max, WithNorm branch (preprocessors.h:623-640) + quantize: w = the input widened to FP32 */
static void quantize_norm(const float *w, const float *mean, size_t dim, int metric, uint8_t *out) {
    float t[dim ? dim : 1];
    float value = w[0] - mean[0];
    float min_val = value, max_val = value;
    float x_mean_ip = w[0] * mean[0];          /* sequential, separate multiply and add (no FMA in this translation unit) */
    t[0] = value;
    for (size_t i = 1; i < dim; i++) {
        value = w[i] - mean[i];
        t[i] = value;
        min_val = (value < min_val) ? value : min_val; /* std::min(min_val, value) */
        max_val = (max_val < value) ? value : max_val; /* std::max(max_val, value) */
        const float p = w[i] * mean[i];
        x_mean_ip += p;
    }
    quantize_core(t, dim, metric, min_val, max_val, metric == VSO_IP ? &x_mean_ip : NULL, out);
}
void vso_sq8_quantize_norm(const float *x, const float *mean, size_t dim, int metric, uint8_t *out) {
    quantize_norm(x, mean, dim, metric, out);
}
void vso_sq8_quantize_norm_f16(const uint16_t *x, const float *mean, size_t dim, int metric, uint8_t *out) {
    float w[dim ? dim : 1];
    for (size_t i = 0; i < dim; i++) w[i] = vso_f16_to_f32(x[i]);
    quantize_norm(w, mean, dim, metric, out);
}

/* assign_query_metadata (preprocessors.h:398-470): v = the query body widened to FP32 (centred for L2), orig = the original
 * input widened; meta = { y_sum, y_sum_squares (L2) | y_mean_ip (IP) } */
static void query_meta_norm(const float *v, const float *orig, const float *mean, size_t dim, int metric, float *meta) {
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0};
    const size_t d4 = dim & ~(size_t)3;
    size_t i = 0;
    for (; i < d4; i += 4)
        for (int j = 0; j < 4; j++) {
            s[j] += v[i + j];
            const float sq = v[i + j] * v[i + j];
            q[j] += sq;
            const float mp = mean[i + j] * orig[i + j];
            m[j] += mp;
        }
    float sum = (s[0] + s[1]) + (s[2] + s[3]);
    float ssq = (q[0] + q[1]) + (q[2] + q[3]);
    float mip = (m[0] + m[1]) + (m[2] + m[3]);
    for (; i < dim; i++) {
        sum += v[i];
        const float sq = v[i] * v[i];
        ssq += sq;
        const float mp = mean[i] * orig[i];
        mip += mp;
    }
    meta[0] = sum;
    meta[1] = metric == VSO_L2 ? ssq : mip;
}
/* preprocessQuery (preprocessors.h:574-598) */
void vso_sq8_query_blob_norm(const float *y, const float *mean, size_t dim, int metric, float *out) {
    float body[dim ? dim : 1];
    for (size_t i = 0; i < dim; i++) body[i] = metric == VSO_L2 ? y[i] - mean[i] : y[i];
    float meta[2];
    query_meta_norm(body, y, mean, dim, metric, meta);
    memcpy(out, body, dim * sizeof(float));
    out[dim] = meta[0];
    out[dim + 1] = meta[1];
}
void vso_sq8_query_blob_norm_f16(const uint16_t *y, const float *mean, size_t dim, int metric, void *out) {
    uint16_t body[dim ? dim : 1];
    float v[dim ? dim : 1], orig[dim ? dim : 1];
    for (size_t i = 0; i < dim; i++) {
        orig[i] = vso_f16_to_f32(y[i]);
        body[i] = metric == VSO_L2 ? vso_f32_to_f16(orig[i] - mean[i]) : y[i]; /* from_fp32(to_fp32(input) - mean) */
        v[i] = vso_f16_to_f32(body[i]);
    }
    float meta[2];
    query_meta_norm(v, orig, mean, dim, metric, meta);
    memcpy(out, body, dim * 2);
    memcpy((char *)out + dim * 2, meta, sizeof meta);
}

/* DistanceCalculatorWithNorm (calculator.h:168-204).  The base kernels read their metadata at the unchanged offsets. */
double vso_sq8_fp32_distance_norm(int metric, int tier, size_t dim, const void *storage, const void *query) {
    const float base = (float)vso_sq8_fp32_distance(metric, tier, dim, storage, query);
    if (metric == VSO_L2) return (double)base;
    const float y_mean_ip = ((const float *)query)[dim + 1];
    return (double)(base - y_mean_ip);
}
double vso_sq8_fp16_distance_norm(int metric, int tier, size_t dim, const void *storage, const void *query) {
    const float base = (float)vso_sq8_fp16_distance(metric, tier, dim, storage, query);
    if (metric == VSO_L2) return (double)base;
    const float y_mean_ip = ldf((const uint8_t *)query + dim * 2 + 4);
    return (double)(base - y_mean_ip);
}
double vso_sq8_sq8_distance_norm(int metric, int tier, size_t dim, const void *a, const void *b, float mean_sum_squares) {
    const float base = (float)vso_sq8_sq8_distance(metric, tier, dim, a, b);
    if (metric == VSO_L2) return (double)base;
    const float x_mean_ip = ldf((const uint8_t *)a + dim + 12), y_mean_ip = ldf((const uint8_t *)b + dim + 12);
    return (double)(((base - x_mean_ip) - y_mean_ip) + mean_sum_squares);
}
void vso_sq8_fp32_scan_norm(int metric, int tier, size_t dim, const void *rows, size_t n, size_t stride, const void *query,
                            double *out) {
    for (size_t i = 0; i < n; i++) out[i] = vso_sq8_fp32_distance_norm(metric, tier, dim, (const char *)rows + i * stride, query);
}
void vso_sq8_fp16_scan_norm(int metric, int tier, size_t dim, const void *rows, size_t n, size_t stride, const void *query,
                            double *out) {
    for (size_t i = 0; i < n; i++) out[i] = vso_sq8_fp16_distance_norm(metric, tier, dim, (const char *)rows + i * stride, query);
}
