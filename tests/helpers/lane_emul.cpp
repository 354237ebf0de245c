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
