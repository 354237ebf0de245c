// blob_prep.h -- host-side blob preprocessing: once per stored vector / per query, O(dim).
//
// Cosine indexes store L2-normalised blobs (fp types) or the raw integers followed by their float
// norm (int8/uint8); the GPU only ever sees such preprocessed blobs.  The arithmetic follows the
// reference exactly because it decides the stored bytes:
//   fp32/fp64  spaces/normalize/normalize_naive.h:24-37   (double sum, norm cast to T, divide)
//   bf16/fp16  :40-78                                       (fp32 sum, fp32 divide, re-round)
//   int8/uint8 :81-88 + compute_norm.h:18-31                (uint64 sum of squares, float norm appended)
// bf16/fp16 rounding helpers restate types/bfloat16.h:23-30 and types/float16.h:62-117.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "VecSim/vec_sim_common.h"

namespace vsa {

inline size_t type_size(VecSimType t) {
    switch (t) {
    case VecSimType_FLOAT32: return 4;
    case VecSimType_FLOAT64: return 8;
    case VecSimType_BFLOAT16:
    case VecSimType_FLOAT16: return 2;
    case VecSimType_INT8:
    case VecSimType_UINT8: return 1;
    case VecSimType_INT32: return 4;
    case VecSimType_INT64: return 8;
    }
    return 0;
}
inline bool is_int_type(VecSimType t) { return t == VecSimType_INT8 || t == VecSimType_UINT8; }

// storedDataSize / query blob size (utils/vec_utils.cpp:296-302, vec_sim.cpp:256-266)
inline size_t blob_bytes(VecSimType t, size_t dim, VecSimMetric m) {
    size_t b = dim * type_size(t);
    if (m == VecSimMetric_Cosine && is_int_type(t)) b += sizeof(float);
    return b;
}

inline float bits_to_f32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t f32_to_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

inline float bf16_widen(uint16_t h) { return bits_to_f32((uint32_t)h << 16); }
inline uint16_t bf16_round(float f) {  // nearest-even on the top 16 bits
    uint32_t u = f32_to_bits(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float fp16_widen(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t em = (uint32_t)(h & 0x7FFFu) << 13;
    uint32_t exp = em & (0x7C00u << 13);
    uint32_t out;
    if (exp == (0x7C00u << 13)) out = em + ((127u - 15u) << 23) + ((128u - 16u) << 23);  // inf / nan
    else if (exp == 0) out = f32_to_bits(bits_to_f32(em + ((127u - 15u) << 23) + (1u << 23)) - bits_to_f32(113u << 23));
    else out = em + ((127u - 15u) << 23);
    return bits_to_f32(out | sign);
}

// Can a score computed from these values be NaN?  Conservative: true for NaN / Inf elements and for magnitudes whose
// products could overflow to +-Inf (and then cancel) in the accumulation type.  With every element of both vectors
// below the bound no partial sum of the reference kernels overflows, so no score is NaN.  (fp16 tops out at 65504.)
inline bool values_may_nan(const void *p, VecSimType t, size_t n) {
    const char *b = static_cast<const char *>(p);
    bool w = false;
    switch (t) {
    case VecSimType_FLOAT32:
        for (size_t i = 0; i < n; i++) {
            float f;
            std::memcpy(&f, b + 4 * i, 4);
            w |= !(std::fabs(f) <= 1e15f);
        }
        return w;
    case VecSimType_FLOAT64:
        for (size_t i = 0; i < n; i++) {
            double f;
            std::memcpy(&f, b + 8 * i, 8);
            w |= !(std::fabs(f) <= 1e150);
        }
        return w;
    case VecSimType_BFLOAT16:
        for (size_t i = 0; i < n; i++) {
            uint16_t h;
            std::memcpy(&h, b + 2 * i, 2);
            w |= !(std::fabs(bf16_widen(h)) <= 1e15f);
        }
        return w;
    case VecSimType_FLOAT16:
        for (size_t i = 0; i < n; i++) {
            uint16_t h;
            std::memcpy(&h, b + 2 * i, 2);
            w |= (h & 0x7C00u) == 0x7C00u;
        }
        return w;
    default: return false;
    }
}
// The reference's narrowing is not plain IEEE round-to-nearest-even: it drops the low 12 mantissa
// bits, rescales by 2^-112 (letting fp32 subnormals model fp16 subnormals), clamps, adds 0x1000
// and shifts.  Stored Cosine fp16 bytes depend on exactly this.
inline uint16_t fp16_round(float f) {
    uint32_t x = f32_to_bits(f);
    const uint32_t sign = x & 0x80000000u;
    x ^= sign;
    const uint32_t inf32 = 255u << 23;
    uint32_t o = (x > inf32) ? 0x7E00u : 0x7C00u;
    if (x < inf32) {
        float scaled = bits_to_f32(x & ~0xFFFu) * bits_to_f32(15u << 23);
        const float cap = bits_to_f32((31u << 23) - 0x1000u);
        if (cap < scaled) scaled = cap;
        int32_t y = (int32_t)(f32_to_bits(scaled) + 0x1000u);
        o = (uint32_t)(y >> 13);
    }
    return (uint16_t)(o | (sign >> 16));
}

// In-place; for int types the buffer must have room for dim + 4 bytes.
inline void normalize_blob(void *blob, size_t dim, VecSimType type) {
    switch (type) {
    case VecSimType_FLOAT32: {
        float *v = static_cast<float *>(blob);
        double ss = 0;
        for (size_t i = 0; i < dim; i++) ss += (double)v[i] * (double)v[i];
        const float norm = (float)std::sqrt(ss);
        for (size_t i = 0; i < dim; i++) v[i] = v[i] / norm;
        break;
    }
    case VecSimType_FLOAT64: {
        double *v = static_cast<double *>(blob);
        double ss = 0;
        for (size_t i = 0; i < dim; i++) ss += v[i] * v[i];
        const double norm = std::sqrt(ss);
        for (size_t i = 0; i < dim; i++) v[i] = v[i] / norm;
        break;
    }
    case VecSimType_BFLOAT16:
    case VecSimType_FLOAT16: {
        uint16_t *v = static_cast<uint16_t *>(blob);
        const bool bf = (type == VecSimType_BFLOAT16);
        std::vector<float> wide(dim);
        volatile float ss = 0;  // volatile: keep the fp32 mul and add separate (no contraction)
        for (size_t i = 0; i < dim; i++) {
            wide[i] = bf ? bf16_widen(v[i]) : fp16_widen(v[i]);
            volatile float sq = wide[i] * wide[i];
            ss = ss + sq;
        }
        const float norm = (float)std::sqrt((double)ss);
        for (size_t i = 0; i < dim; i++) {
            const float q = wide[i] / norm;
            v[i] = bf ? bf16_round(q) : fp16_round(q);
        }
        break;
    }
    case VecSimType_INT8:
    case VecSimType_UINT8: {
        uint64_t ss = 0;
        if (type == VecSimType_INT8) {
            const int8_t *v = static_cast<const int8_t *>(blob);
            for (size_t i = 0; i < dim; i++) ss += (uint64_t)((int)v[i] * (int)v[i]);
        } else {
            const uint8_t *v = static_cast<const uint8_t *>(blob);
            for (size_t i = 0; i < dim; i++) ss += (uint64_t)((int)v[i] * (int)v[i]);
        }
        const float norm = (float)std::sqrt((double)ss);
        std::memcpy(static_cast<char *>(blob) + dim, &norm, sizeof norm);
        break;
    }
    default: break;
    }
}

}  // namespace vsa
