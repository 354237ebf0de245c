// lane_emul.cpp -- TEST HELPER (not part of the product libraries).
// Walks a lane program (vectorsimilarity_amd/csrc/lane_program.h) on the host exactly the way
// k_exact_scan walks it on the GPU -- one accumulator per virtual lane, steps in order, then the
// halving tree with offsets vl/2..1 -- so the CPU test-suite can check the *tables* against the
// oracle without a GPU.  Built on demand by tests/test_lane_program.py with g++ -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "lane_program.h"

static float widen16(int type, uint16_t h) {
    if (type == VSGPU_BF16) {
        uint32_t u = (uint32_t)h << 16;
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    }
    // exact fp16 -> fp32 widening
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, u;
    if (e == 0x1F) u = sign | 0x7F800000u | (m << 13);
    else if (e != 0) u = sign | ((e + 112u) << 23) | (m << 13);
    else if (m == 0) u = sign;
    else {
        int sh = 0;
        while (!(m & 0x400u)) { m <<= 1; sh++; }
        u = sign | ((uint32_t)(113 - sh) << 23) | ((m & 0x3FFu) << 13);
    }
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

template <typename T> static T tree(std::vector<T> v, int vl) {
    for (int o = vl / 2; o >= 1; o >>= 1)
        for (int i = 0; i < o; i++) v[i] = v[i] + v[i + o];
    return v[0];
}

// the F16C kernel's horizontal add (LaneProgram::reduce == 1)
static float reduce_f16c(const std::vector<float> &v) {
    float t[8];
    for (int j = 0; j < 8; j++) t[j] = (v[j] + v[j + 8]) + 0.0f;
    float r = t[0];
    for (int j = 1; j < 8; j++) r = r + t[j];
    return r;
}

extern "C" int lane_emul_steps(int type, int metric, int tier, size_t dim) {
    return vsg::build_lane_program(type, metric, tier, dim).steps;
}

// returns the accumulator after the tree (before the 1-x / int epilogue), as double
extern "C" double lane_emul(int type, int metric, int tier, size_t dim, const void *a, const void *b) {
    vsg::LaneProgram p = vsg::build_lane_program(type, metric, tier, dim);
    const char *pa = (const char *)a, *pb = (const char *)b;
    if (type == VSGPU_F64) {
        std::vector<double> acc(p.vl, 0.0);
        for (int s = 0; s < p.steps; s++)
            for (int l = 0; l < p.vl; l++) {
                int off = p.offs[(size_t)s * p.vl + l];
                if (off < 0) continue;
                double x, q;
                std::memcpy(&x, pa + off, 8);
                std::memcpy(&q, pb + off, 8);
                if (p.is_l2) {
                    double t = x - q;
                    acc[l] = p.fused ? std::fma(t, t, acc[l]) : acc[l] + t * t;
                } else {
                    acc[l] = p.fused ? std::fma(x, q, acc[l]) : acc[l] + x * q;
                }
            }
        return tree(acc, p.vl);
    }
    if (type == VSGPU_I8 || type == VSGPU_U8) {
        std::vector<long long> acc(p.vl, 0);
        for (int s = 0; s < p.steps; s++)
            for (int l = 0; l < p.vl; l++) {
                int off = p.offs[(size_t)s * p.vl + l];
                if (off < 0) continue;
                int x = type == VSGPU_I8 ? (int)*(const int8_t *)(pa + off) : (int)*(const uint8_t *)(pa + off);
                int q = type == VSGPU_I8 ? (int)*(const int8_t *)(pb + off) : (int)*(const uint8_t *)(pb + off);
                acc[l] += p.is_l2 ? (long long)(x - q) * (x - q) : (long long)x * q;
            }
        return (double)tree(acc, p.vl);
    }
    std::vector<float> acc(p.vl, 0.0f);
    for (int s = 0; s < p.steps; s++)
        for (int l = 0; l < p.vl; l++) {
            int off = p.offs[(size_t)s * p.vl + l];
            if (off < 0) continue;
            float x, q;
            if (type == VSGPU_F32) {
                std::memcpy(&x, pa + off, 4);
                std::memcpy(&q, pb + off, 4);
            } else {
                uint16_t hx, hq;
                std::memcpy(&hx, pa + off, 2);
                std::memcpy(&hq, pb + off, 2);
                x = widen16(type, hx);
                q = widen16(type, hq);
            }
            if (p.dpbf16) {
                auto ftz = [](float v) {
                    uint32_t u;
                    std::memcpy(&u, &v, 4);
                    if ((u & 0x7f800000u) == 0) u &= 0x80000000u;
                    std::memcpy(&v, &u, 4);
                    return v;
                };
                acc[l] = ftz(std::fma(ftz(x), ftz(q), ftz(acc[l])));
            } else if (p.is_l2) {
                float t = x - q;
                if (p.fused) acc[l] = std::fma(t, t, acc[l]);
                else { float m = t * t; acc[l] = acc[l] + m; }
            } else {
                if (p.fused) acc[l] = std::fma(x, q, acc[l]);
                else { float m = x * q; acc[l] = acc[l] + m; }
            }
        }
    if (p.reduce == 1) return (double)reduce_f16c(acc);
    return (double)tree(acc, p.vl);
}

// SQ8 storage x FP32 query (VSGPU_SQ8): walks the table like k_exact_scan<EK_SQ8> (float(code) * y, fp32 accumulators, the
// halving tree) and applies the epilogue the way exact_kernels.hpp:sq8_score does.  metric: 0 L2, 1 IP / Cosine.
extern "C" double lane_emul_sq8(int metric, int tier, size_t dim, const void *storage, const void *query_blob) {
    vsg::LaneProgram p = vsg::build_lane_program(VSGPU_SQ8, VSGPU_IP, tier, dim);
    const unsigned char *c = (const unsigned char *)storage;
    const char *y = (const char *)query_blob;
    std::vector<float> acc(p.vl, 0.0f);
    for (int s = 0; s < p.steps; s++)
        for (int l = 0; l < p.vl; l++) {
            int off = p.offs[(size_t)s * p.vl + l];
            if (off < 0) continue;
            const float x = (float)c[off];
            float q;
            std::memcpy(&q, y + 4 * (size_t)off, 4);
            if (p.fused) acc[l] = std::fma(x, q, acc[l]);
            else { float m = x * q; acc[l] = acc[l] + m; }
        }
    const float qdot = tree(acc, p.vl);
    float meta[4] = {0, 0, 0, 0}, qm[2] = {0, 0};
    std::memcpy(meta, c + dim, metric == 0 ? 16 : 12);
    std::memcpy(qm, y + 4 * dim, metric == 0 ? 8 : 4);
    const float dq = meta[1] * qdot;
    float ip;
    if (p.fused) ip = std::fma(meta[0], qm[0], dq);
    else { float a = meta[0] * qm[0]; ip = a + dq; }
    if (metric != 0) return (double)(1.0f - ip);
    const float t = meta[3] + qm[1];
    const float two_ip = 2.0f * ip;
    return (double)(t - two_ip);
}

// SQ8 storage x FP16 query (VSGPU_SQ8H): 64 virtual lanes, fp16 query values widened exactly, metadata behind 2 dim bytes
extern "C" double lane_emul_sq8h(int metric, int tier, size_t dim, const void *storage, const void *query_blob) {
    vsg::LaneProgram p = vsg::build_lane_program(VSGPU_SQ8H, VSGPU_IP, tier, dim);
    const unsigned char *c = (const unsigned char *)storage;
    const char *y = (const char *)query_blob;
    std::vector<float> acc(p.vl, 0.0f);
    for (int s = 0; s < p.steps; s++)
        for (int l = 0; l < p.vl; l++) {
            int off = p.offs[(size_t)s * p.vl + l];
            if (off < 0) continue;
            const float x = (float)c[off];
            uint16_t h;
            std::memcpy(&h, y + 2 * (size_t)off, 2);
            const float q = widen16(VSGPU_F16, h);
            if (p.fused) acc[l] = std::fma(x, q, acc[l]);
            else { float m = x * q; acc[l] = acc[l] + m; }
        }
    const float qdot = tree(acc, p.vl);
    float meta[4] = {0, 0, 0, 0}, qm[2] = {0, 0};
    std::memcpy(meta, c + dim, metric == 0 ? 16 : 12);
    std::memcpy(qm, y + 2 * dim, metric == 0 ? 8 : 4);
    const float dq = meta[1] * qdot;
    float ip;
    if (p.fused) ip = std::fma(meta[0], qm[0], dq);
    else { float a = meta[0] * qm[0]; ip = a + dq; }
    if (metric != 0) return (double)(1.0f - ip);
    const float t = meta[3] + qm[1];
    const float two_ip = 2.0f * ip;
    return (double)(t - two_ip);
}
