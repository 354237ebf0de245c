// sq8_prep.h -- host-side SQ8 blob preprocessing: once per stored vector / per query, O(dim).
//
// Restates QuantPreprocessor<float, Metric, WithNorm = false> (spaces/computer/preprocessors.h:259-649), because it
// decides the stored bytes and the query metadata the kernels consume:
//   storage  | codes[dim] u8 | min | delta | sum | sum_squares (L2 only) |   quantize()               :270-390
//   query    | y[dim] f32    | y_sum | y_sum_squares (L2 only) |            assign_query_metadata()  :398-470
// (types/sq8.h:19-62 for the layouts).  Cosine indexes normalise the fp32 vector first (blob_prep.h), then quantise.
// FP16 inputs (QuantPreprocessor<float16, ...>): every value is widened exactly and all arithmetic is FP32, so storage
// blobs are the FP32 quantiser's on the widened vector; a query blob keeps the fp16 values and appends FP32 metadata (at an
// offset that is not 4-byte aligned in general).
// Mean-centred blobs (WithNorm = true, L2 and IP; preprocessors.h:484-495, 574-640, types/sq8.h:38-58): storage = SQ8 of
// x - mean, IP rows carry x_mean_ip = sum x_i mean_i behind the three base slots; L2 queries are stored centred, IP queries
// raw with y_mean_ip = sum mean_i y_i behind y_sum.  The distance on top (DistanceCalculatorWithNorm, calculator.h:168-204)
// is the base kernel's, minus y_mean_ip for IP.
#pragma once
#include <cstdint>
#include <cstring>

#include "VecSim/vec_sim_common.h"

namespace vsa {

inline size_t sq8_storage_bytes(size_t dim, VecSimMetric m) { return dim + (m == VecSimMetric_L2 ? 4 : 3) * sizeof(float); }
inline size_t sq8_query_bytes(size_t dim, VecSimMetric m) { return (dim + (m == VecSimMetric_L2 ? 2 : 1)) * sizeof(float); }
inline size_t sq8_query_bytes_f16(size_t dim, VecSimMetric m) { return dim * 2 + (m == VecSimMetric_L2 ? 2 : 1) * sizeof(float); }
// mean-centred (WithNorm) blobs: one more FP32 slot on either side for IP, nothing for L2
inline size_t sq8_storage_bytes(size_t dim, VecSimMetric m, bool centred) {
    return sq8_storage_bytes(dim, m) + ((centred && m == VecSimMetric_IP) ? sizeof(float) : 0);
}
inline size_t sq8_query_bytes(size_t dim, VecSimMetric m, bool centred, bool f16) {
    return (f16 ? sq8_query_bytes_f16(dim, m) : sq8_query_bytes(dim, m)) + ((centred && m == VecSimMetric_IP) ? sizeof(float) : 0);
}

// bounded conversion, then +0.5 and truncate (preprocessors.h:287-299): zero / negative / NaN -> 0, >= 255 / +inf -> 255
inline uint8_t sq8_to_byte(float scaled) {
    if (!(scaled > 0.0f)) return 0;
    if (scaled >= 255.0f) return 255;
    return (uint8_t)(scaled + 0.5f);
}

// quantize() after find_min_max: t = transformed_value() of every element (the input, or input - mean); `tail` = the extra
// x_mean_ip slot of mean-centred IP blobs
inline void sq8_quantize_core(const float *t, size_t dim, VecSimMetric metric, float min_val, float max_val, const float *tail,
                              uint8_t *out) {
    const float diff = max_val - min_val;
    const float delta = (diff == 0.0f) ? 1.0f : diff / 255.0f;
    const float inv_delta = 1.0f / delta;
    // byte sums as exact integers (four chains like the reference; integer sums do not depend on the order)
    uint32_t q_sum = 0;
    uint64_t q_sq = 0;
    for (size_t i = 0; i < dim; i++) {
        const uint8_t a = sq8_to_byte((t[i] - min_val) * inv_delta);
        out[i] = a;
        q_sum += a;
        q_sq += (uint64_t)a * a;
    }
    // sums of the reconstruction min + delta * a[i], expanded in double (:369-381), stored as FP32
    const double d_min = min_val, d_delta = delta, d_dim = (double)dim;
    float meta[5] = {min_val, delta, (float)(d_dim * d_min + d_delta * (double)q_sum), 0.0f, 0.0f};
    size_t n = 3;
    if (metric == VecSimMetric_L2) {
        const double t0 = d_dim * d_min * d_min, t1 = 2.0 * d_min * d_delta * (double)q_sum, t2 = d_delta * d_delta * (double)q_sq;
        meta[n++] = (float)((t0 + t1) + t2);
    }
    if (tail) meta[n++] = *tail;
    std::memcpy(out + dim, meta, n * sizeof(float));   // the metadata offset is not 4-byte aligned in general
}
inline void sq8_quantize(const float *x, size_t dim, VecSimMetric metric, uint8_t *out) {
    // std::minmax_element: the first smallest, the last largest (:620-622)
    float min_val = x[0], max_val = x[0];
    for (size_t i = 1; i < dim; i++) {
        if (x[i] < min_val) min_val = x[i];
        if (!(x[i] < max_val)) max_val = x[i];
    }
    sq8_quantize_core(x, dim, metric, min_val, max_val, nullptr, out);
}
// WithNorm: x = the input widened to FP32; find_min_max's second branch (:623-640) runs std::min / std::max over x - mean
// and accumulates x_mean_ip left to right
inline void sq8_quantize_centred(const float *x, const float *mean, size_t dim, VecSimMetric metric, uint8_t *out, float *scratch) {
    float value = x[0] - mean[0];
    float min_val = value, max_val = value, x_mean_ip = x[0] * mean[0];
    scratch[0] = value;
    for (size_t i = 1; i < dim; i++) {
        value = x[i] - mean[i];
        scratch[i] = value;
        min_val = (value < min_val) ? value : min_val;
        max_val = (max_val < value) ? value : max_val;
        x_mean_ip += x[i] * mean[i];
    }
    sq8_quantize_core(scratch, dim, metric, min_val, max_val, metric == VecSimMetric_IP ? &x_mean_ip : nullptr, out);
}
// WithNorm query metadata (assign_query_metadata, :398-470): v = the query body widened to FP32 (centred for L2), orig = the
// original input widened; meta[1] = y_sum_squares (L2) or y_mean_ip (IP)
inline void sq8_query_meta_centred(const float *v, const float *orig, const float *mean, size_t dim, VecSimMetric metric, float meta[2]) {
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0};
    const size_t d4 = dim & ~(size_t)3;
    size_t i = 0;
    for (; i < d4; i += 4)
        for (int j = 0; j < 4; j++) {
            s[j] += v[i + j];
            q[j] += v[i + j] * v[i + j];
            m[j] += mean[i + j] * orig[i + j];
        }
    float sum = (s[0] + s[1]) + (s[2] + s[3]), sq = (q[0] + q[1]) + (q[2] + q[3]), mip = (m[0] + m[1]) + (m[2] + m[3]);
    for (; i < dim; i++) {
        sum += v[i];
        sq += v[i] * v[i];
        mip += mean[i] * orig[i];
    }
    meta[0] = sum;
    meta[1] = metric == VecSimMetric_L2 ? sq : mip;
}

// query values followed by y_sum (and y_sum_squares for L2): four fp32 chains, (s0 + s1) + (s2 + s3), tail added after
inline void sq8_query_blob(const float *y, size_t dim, VecSimMetric metric, float *out) {
    if (out != y) std::memmove(out, y, dim * sizeof(float));
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const size_t d4 = dim & ~(size_t)3;
    size_t i = 0;
    for (; i < d4; i += 4)
        for (int j = 0; j < 4; j++) {
            const float v = out[i + j];
            s[j] += v;
            q[j] += v * v;
        }
    float sum = (s[0] + s[1]) + (s[2] + s[3]);
    float sq = (q[0] + q[1]) + (q[2] + q[3]);
    for (; i < dim; i++) {
        sum += out[i];
        sq += out[i] * out[i];
    }
    out[dim] = sum;
    if (metric == VecSimMetric_L2) out[dim + 1] = sq;
}

// fp16 query blob: `blob` holds dim fp16 values (already normalised for Cosine) and room for the metadata behind them;
// `wide` = the same values widened to FP32
inline void sq8_query_meta_f16(const float *wide, size_t dim, VecSimMetric metric, char *blob) {
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const size_t d4 = dim & ~(size_t)3;
    size_t i = 0;
    for (; i < d4; i += 4)
        for (int j = 0; j < 4; j++) {
            s[j] += wide[i + j];
            q[j] += wide[i + j] * wide[i + j];
        }
    float meta[2] = {(s[0] + s[1]) + (s[2] + s[3]), (q[0] + q[1]) + (q[2] + q[3])};
    for (; i < dim; i++) {
        meta[0] += wide[i];
        meta[1] += wide[i] * wide[i];
    }
    std::memcpy(blob + dim * 2, meta, (metric == VecSimMetric_L2 ? 2 : 1) * sizeof(float));
}

}  // namespace vsa
