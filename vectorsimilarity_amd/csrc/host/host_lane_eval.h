// host_lane_eval.h -- the reference-order distance of two STORED rows on the host, by walking the same lane program
// (csrc/lane_program.h) the exact GPU kernels walk: one accumulator per virtual lane, steps in order, the kernel's own
// horizontal add, the kernel's own epilogue (exact_kernels.hpp epilogue_score).
//
// INGEST ONLY.  The single user is the HNSW index's reference-order insert path (hnsw_ref_build.cpp): the reference's graph is
// a function of its build-time distances (hnsw.h:1567-1610 calls the index's dist_func for every candidate), so an index that
// is to equal the reference's after the same AddVector calls has to rank candidates by the same numbers.  No query entry point
// reaches this file; every distance a query needs is evaluated by the gfx950 kernels.
//
// Rows are read from the index's host copies: the stored blob (`raw`) and -- for fp32 / bf16 / fp16 rows -- its exact fp32
// widening (`wide`; fp32 rows: the same values).  Runs of steps in which every lane reads the next consecutive element (the
// body of the two-accumulator AVX-512 shape) go through a vector loop when the host has AVX-512F; every other step (masked
// heads, the bf16 lane shuffles, the scalar tier's single chain, vdpbf16ps with its flushes) goes lane by lane.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "lane_program.h"

namespace vsa {

class HostLaneEval {
public:
    // type: VSGPU_F32 ... VSGPU_U8; metric: the INDEX metric (VSGPU_L2 / IP / COSINE); false: this (type, tier) has no host walker
    bool init(int type, int metric, int tier, size_t dim) {
        type_ = type;
        dim_ = dim;
        const bool is_int = type == VSGPU_I8 || type == VSGPU_U8;
        if (type < VSGPU_F32 || type > VSGPU_U8) return false;
        l2_ = metric == VSGPU_L2;
        int_cos_ = is_int && metric == VSGPU_COSINE;
        int_ip_ = is_int && metric == VSGPU_IP;
        prog_ = vsg::build_lane_program(type, l2_ ? VSGPU_L2 : VSGPU_IP, tier, dim);
        if (prog_.f16acc || prog_.reduce == 2) return false;   // AVX512-FP16 tier: half accumulators, exact GPU kernels only
        const int vl = prog_.vl, eb = prog_.elem_bytes;
        elem_.assign(prog_.offs.size(), -1);
        for (size_t i = 0; i < prog_.offs.size(); i++)
            if (prog_.offs[i] >= 0) elem_[i] = prog_.offs[i] / eb;
        // runs of consecutive steps whose lane l reads element base + l, base advancing by vl per step
        runs_.clear();
        step_run_.assign((size_t)prog_.steps, -1);
        if (vl == 32 && prog_.fused && !prog_.dpbf16 && (type == VSGPU_F32 || type == VSGPU_F16) && have_avx512()) {
            for (int s = 0; s < prog_.steps;) {
                if (!contiguous(s)) {
                    s++;
                    continue;
                }
                int e = s + 1;
                while (e < prog_.steps && contiguous(e) && elem_[(size_t)e * vl] == elem_[(size_t)(e - 1) * vl] + vl) e++;
                for (int k = s; k < e; k++) step_run_[(size_t)k] = (int)runs_.size();
                runs_.push_back(Run{s, e - s, elem_[(size_t)s * vl]});
                s = e;
            }
        }
        ok_ = true;
        return true;
    }
    bool ok() const { return ok_; }
    // a, b: stored blobs; wa, wb: their fp32 images (fp32 / bf16 / fp16 rows; may be null otherwise).  The value the reference's
    // dist_func returns for the pair (float-valued but for fp64), as a double.
    double score(const char *a, const char *b, const float *wa, const float *wb) const {
        if (type_ == VSGPU_I8 || type_ == VSGPU_U8) return int_score(a, b);
        if (type_ == VSGPU_F64) return f64_score(a, b);
        return (double)f32_score(wa, wb);
    }

private:
    struct Run {
        int first_step, n_steps, first_elem;
    };
    static bool have_avx512() {
#if defined(__x86_64__)
        static const bool v = (__builtin_cpu_init(), __builtin_cpu_supports("avx512f"));
        return v;
#else
        return false;
#endif
    }
    bool contiguous(int s) const {
        const int vl = prog_.vl;
        const int32_t base = elem_[(size_t)s * vl];
        if (base < 0) return false;
        for (int l = 1; l < vl; l++)
            if (elem_[(size_t)s * vl + l] != base + l) return false;
        return true;
    }
    static float ftz(float v) {
        uint32_t u;
        std::memcpy(&u, &v, 4);
        if ((u & 0x7f800000u) == 0) u &= 0x80000000u;
        std::memcpy(&v, &u, 4);
        return v;
    }
#if defined(__x86_64__)
    // `n` consecutive 32-element steps: lanes 0-15 -> acc[0..15], lanes 16-31 -> acc[16..31] (one fma per lane and step, in step order)
    __attribute__((target("avx512f"))) static void run32_avx512(const float *x, const float *q, int n, bool l2, float *acc) {
        __m512 a0 = _mm512_loadu_ps(acc), a1 = _mm512_loadu_ps(acc + 16);
        if (l2) {
            for (int i = 0; i < n; i++, x += 32, q += 32) {
                const __m512 t0 = _mm512_sub_ps(_mm512_loadu_ps(x), _mm512_loadu_ps(q));
                const __m512 t1 = _mm512_sub_ps(_mm512_loadu_ps(x + 16), _mm512_loadu_ps(q + 16));
                a0 = _mm512_fmadd_ps(t0, t0, a0);
                a1 = _mm512_fmadd_ps(t1, t1, a1);
            }
        } else {
            for (int i = 0; i < n; i++, x += 32, q += 32) {
                a0 = _mm512_fmadd_ps(_mm512_loadu_ps(x), _mm512_loadu_ps(q), a0);
                a1 = _mm512_fmadd_ps(_mm512_loadu_ps(x + 16), _mm512_loadu_ps(q + 16), a1);
            }
        }
        _mm512_storeu_ps(acc, a0);
        _mm512_storeu_ps(acc + 16, a1);
    }
#endif
    float f32_score(const float *x, const float *q) const {
        const int vl = prog_.vl;
        float acc[64];
        for (int l = 0; l < vl; l++) acc[l] = 0.0f;
        for (int s = 0; s < prog_.steps;) {
#if defined(__x86_64__)
            if (step_run_[(size_t)s] >= 0) {
                const Run &r = runs_[(size_t)step_run_[(size_t)s]];
                run32_avx512(x + r.first_elem, q + r.first_elem, r.n_steps, l2_, acc);
                s = r.first_step + r.n_steps;
                continue;
            }
#endif
            const int32_t *e = &elem_[(size_t)s * vl];
            for (int l = 0; l < vl; l++) {
                if (e[l] < 0) continue;
                const float xv = x[e[l]], qv = q[e[l]];
                if (prog_.dpbf16) {
                    acc[l] = ftz(std::fma(ftz(xv), ftz(qv), ftz(acc[l])));
                } else if (l2_) {
                    const float t = xv - qv;
                    if (prog_.fused) acc[l] = std::fma(t, t, acc[l]);
                    else {
                        const float m = t * t;
                        acc[l] = acc[l] + m;
                    }
                } else {
                    if (prog_.fused) acc[l] = std::fma(xv, qv, acc[l]);
                    else {
                        const float m = xv * qv;
                        acc[l] = acc[l] + m;
                    }
                }
            }
            s++;
        }
        float tot;
        if (prog_.reduce == 1) {   // the F16C kernel's horizontal add (AVX_utils.h:32-37)
            float t[8];
            for (int j = 0; j < 8; j++) t[j] = (acc[j] + acc[j + 8]) + 0.0f;
            tot = t[0];
            for (int j = 1; j < 8; j++) tot = tot + t[j];
        } else {
            for (int o = vl / 2; o >= 1; o >>= 1)
                for (int i = 0; i < o; i++) acc[i] = acc[i] + acc[i + o];
            tot = acc[0];
        }
        return l2_ ? tot : 1.0f - tot;
    }
    double f64_score(const char *a, const char *b) const {
        const int vl = prog_.vl;
        double acc[32];
        for (int l = 0; l < vl; l++) acc[l] = 0.0;
        for (int s = 0; s < prog_.steps; s++) {
            const int32_t *e = &elem_[(size_t)s * vl];
            for (int l = 0; l < vl; l++) {
                if (e[l] < 0) continue;
                double xv, qv;
                std::memcpy(&xv, a + 8 * (size_t)e[l], 8);
                std::memcpy(&qv, b + 8 * (size_t)e[l], 8);
                if (l2_) {
                    const double t = xv - qv;
                    if (prog_.fused) acc[l] = std::fma(t, t, acc[l]);
                    else {
                        const double m = t * t;
                        acc[l] = acc[l] + m;
                    }
                } else {
                    if (prog_.fused) acc[l] = std::fma(xv, qv, acc[l]);
                    else {
                        const double m = xv * qv;
                        acc[l] = acc[l] + m;
                    }
                }
            }
        }
        for (int o = vl / 2; o >= 1; o >>= 1)
            for (int i = 0; i < o; i++) acc[i] = acc[i] + acc[i + o];
        return l2_ ? acc[0] : 1.0 - acc[0];
    }
    // exact integer sums (any order gives the reference's integer); epilogues of exact_kernels.hpp epilogue_score(long long ...)
    double int_score(const char *a, const char *b) const {
        long long acc = 0;
        if (type_ == VSGPU_I8) {
            const int8_t *x = (const int8_t *)a, *q = (const int8_t *)b;
            if (l2_)
                for (size_t i = 0; i < dim_; i++) {
                    const int t = (int)x[i] - (int)q[i];
                    acc += t * t;
                }
            else
                for (size_t i = 0; i < dim_; i++) acc += (int)x[i] * (int)q[i];
        } else {
            const uint8_t *x = (const uint8_t *)a, *q = (const uint8_t *)b;
            if (l2_)
                for (size_t i = 0; i < dim_; i++) {
                    const int t = (int)x[i] - (int)q[i];
                    acc += t * t;
                }
            else
                for (size_t i = 0; i < dim_; i++) acc += (int)x[i] * (int)q[i];
        }
        if (l2_) return (double)(float)acc;                       // L2.cpp:164-174
        if (int_ip_) return (double)(float)(1ll - acc);           // IP.cpp:258-262
        float na, nb;                                             // IP.cpp:264-271: norms behind the elements
        std::memcpy(&na, a + dim_, 4);
        std::memcpy(&nb, b + dim_, 4);
        const float ip = (float)acc;
        const float nn = na * nb;
        const float r = ip / nn;
        return (double)(1.0f - r);
    }

    vsg::LaneProgram prog_;
    std::vector<int32_t> elem_;
    std::vector<int> step_run_;
    std::vector<Run> runs_;
    int type_ = 0;
    size_t dim_ = 0;
    bool l2_ = true, int_cos_ = false, int_ip_ = false, ok_ = false;
};

}  // namespace vsa
