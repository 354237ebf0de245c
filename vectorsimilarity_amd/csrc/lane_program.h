// lane_program.h -- host-side description of the summation order each kernel must reproduce.
//
// The reference picks one SIMD kernel per (type, metric, dim) (spaces/L2_space.cpp:185-516,
// spaces/IP_space.cpp:435-889) and every such kernel is a set of independent accumulator lanes,
// each consuming a fixed subsequence of the vector elements in increasing order, followed by a
// fixed halving-tree horizontal add.  A "lane program" writes that down as a table:
//
//     offs[step * vl + lane] = byte offset (inside the row) of the element that virtual lane `lane`
//                              consumes at `step`, or -1 when the lane idles in that step.
//
// The exact-distance HIP kernels are table driven: one GPU lane plays one virtual lane, walks the
// steps in order, then the group reduces with __shfl_down offsets vl/2 .. 1, which is the
// _mm512_reduce_add_ps / _pd tree of gcc 11 (avx512fintrin.h:16112-16121).  Host code only; no
// distance is ever computed here.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "vsgpu.h"

namespace vsg {

struct LaneProgram {
    int vl = 32;        // virtual lanes per row (32: fp32/fp16/int8/uint8/SQ8, 16: fp64/bf16, 64: SQ8 with fp16 queries)
    int steps = 0;      // table height
    bool fused = true;  // true: acc = fma(a,b,acc); false: acc = acc + a*b (scalar tier, L2.cpp:76-133)
    bool is_l2 = true;  // L2: (x-q)^2 terms; otherwise x*q terms
    int elem_bytes = 4;
    bool scalar_tier = false;
    // how the vl accumulator lanes become one number: 0 = halving tree (offsets vl/2 .. 1), 1 = the F16C kernel's
    // (lane j + lane j+8) + 0, j < 8, then the eight sums added left to right (AVX_utils.h:32-37), 2 = the halving tree in HALF
    // precision (_mm512_reduce_add_ph)
    int reduce = 0;
    // vdpbf16ps step (avx512_bf16 tier, bf16 IP): inputs and result of every fma flushed to zero when subnormal
    bool dpbf16 = false;
    // AVX512-FP16 tier, fp16 rows: ONE 32-lane accumulator of HALF precision (vfmadd...ph; L2: vsubph first), reduced by the
    // halving tree in half precision (reduce = 2), IP finished by a half-precision 1 - x
    bool f16acc = false;
    std::vector<int32_t> offs;  // steps * vl
    // [full_from, full_to): the longest run of steps in which every lane is active (residual handling makes
    // the first step(s) partial for float kernels, the last one partial for the integer striding)
    bool step_full(int s) const {
        for (int l = 0; l < vl; l++)
            if (offs[(size_t)s * vl + l] < 0) return false;
        return true;
    }
    int full_from() const {
        int s = 0;
        while (s < steps && !step_full(s)) s++;
        return s;
    }
    int full_to() const {
        int s = full_from();
        while (s < steps && step_full(s)) s++;
        return s;
    }
};

inline int elem_bytes_of(int type) {
    switch (type) {
    case VSGPU_F32: return 4;
    case VSGPU_F64: return 8;
    case VSGPU_BF16:
    case VSGPU_F16: return 2;
    default: return 1;  // int8, uint8, SQ8 codes
    }
}

// Minimum dims below which the x86 choosers keep the scalar kernel:
// fp32 <8 (L2_space.cpp:215-217, IP_space.cpp twin), fp64 <4 (:274-276), bf16 <32 (:329-331),
// fp16 <8; fp16 dims 8..15 are the F16C tier (:404-409), 16 and up the AVX512F tier (:397-402).
inline bool uses_scalar_tier(int type, int tier, size_t dim) {
    if (tier == VSGPU_TIER_SCALAR) return true;
    switch (type) {
    case VSGPU_F32: return dim < 8;
    case VSGPU_F64: return dim < 4;
    case VSGPU_BF16: return dim < 32;
    case VSGPU_F16: return dim < 8;
    case VSGPU_SQ8: return dim < 8;  // L2_space.cpp:71-75, IP_space.cpp:72-76
    case VSGPU_SQ8H: return dim < 16;  // L2_space.cpp:121-123, IP_space.cpp:191
    default: return false;  // integer kernels are exact in any order
    }
}

inline LaneProgram build_lane_program(int type, int kernel_metric, int tier, size_t dim) {
    LaneProgram p;
    p.elem_bytes = elem_bytes_of(type);
    p.is_l2 = (kernel_metric == VSGPU_L2);
    p.vl = (type == VSGPU_F64 || type == VSGPU_BF16) ? 16 : (type == VSGPU_SQ8H ? 64 : 32);
    const int vl = p.vl;
    const int eb = p.elem_bytes;
    auto new_step = [&]() {
        p.offs.insert(p.offs.end(), (size_t)vl, -1);
        return p.steps++;
    };
    auto put = [&](int step, int lane, size_t elem) {
        p.offs[(size_t)step * vl + lane] = (int32_t)(elem * eb);
    };

    if (type == VSGPU_I8 || type == VSGPU_U8) {
        // exact integer sums: any order gives the reference's integer (IP_AVX512F_BW_VL_VNNI_INT8.h
        // :29-60 accumulates in int32 lanes; L2.cpp:149-162 in long long).  Plain striding.
        p.fused = true;
        for (size_t e = 0; e < dim; e += vl) {
            int s = new_step();
            for (int j = 0; j < vl && e + j < dim; j++) put(s, j, e + j);
        }
        return p;
    }

    if (type == VSGPU_SQ8H && !uses_scalar_tier(type, tier, dim)) {
        // SQ8 x FP16 AVX-512F kernel (IP_AVX512F_SQ8_FP16.h:42-101): FOUR 16-lane accumulators.  The dim % 16 head is a
        // masked multiply into sum0, then 64 elements per round over sum0..sum3, up to three 16-chunks of tail into sum0,
        // sum1, sum2, and (sum0 + sum1) + (sum2 + sum3) before the 16-lane tree.  The 64-lane halving tree adds lanes
        // j + 32 and then j + 16, so sum0, sum2, sum1, sum3 sit on lanes 0-15, 16-31, 32-47, 48-63.
        static const int base_of[4] = {0, 32, 16, 48};
        const size_t residual = dim % 16;
        size_t pos = 0;
        if (residual) {
            int s = new_step();
            for (size_t j = 0; j < residual; j++) put(s, (int)j, j);
            pos = residual;
        }
        while (dim - pos >= 64) {
            int s = new_step();
            for (int a = 0; a < 4; a++, pos += 16)
                for (size_t j = 0; j < 16; j++) put(s, base_of[a] + (int)j, pos + j);
        }
        if (dim - pos >= 16) {
            int s = new_step();
            for (int a = 0; a < 3 && dim - pos >= 16; a++, pos += 16)
                for (size_t j = 0; j < 16; j++) put(s, base_of[a] + (int)j, pos + j);
        }
        return p;
    }
    if ((type == VSGPU_SQ8 || type == VSGPU_SQ8H) && uses_scalar_tier(type, tier, dim)) {
        // SQ8_FP32_InnerProduct_Impl (IP.cpp:34-58): four chains over elements i % 4, the dim % 4 tail into chain 0,
        // separate multiply and add, then (s0 + s1) + (s2 + s3).  The halving tree adds lane 0 + lane 2 and lane 1 +
        // lane 3 before the last step, so chains 0, 1, 2, 3 sit on lanes 0, 2, 1, 3 (the other lanes hold +0).
        static const int lane_of[4] = {0, 2, 1, 3};
        p.fused = false;
        p.scalar_tier = true;
        const size_t d4 = dim & ~(size_t)3;
        for (size_t e = 0; e < d4; e += 4) {
            int s = new_step();
            for (int j = 0; j < 4; j++) put(s, lane_of[j], e + j);
        }
        for (size_t e = d4; e < dim; e++) put(new_step(), 0, e);
        return p;
    }
    if (uses_scalar_tier(type, tier, dim)) {
        // one sequential chain, separate multiply and add
        p.fused = false;
        p.scalar_tier = true;
        for (size_t e = 0; e < dim; e++) put(new_step(), 0, e);
        return p;
    }

    if (type == VSGPU_F16 && tier == VSGPU_TIER_AVX512_FP16 && dim >= 32) {
        // IP_AVX512FP16_VL_FP16.h:27-51, L2_AVX512FP16_VL_FP16.h:29-58 (choosers IP_space.cpp:649-658, L2_space.cpp:388-397):
        // the dim % 32 head through a zero mask (a multiply there; an fma onto +0 gives the same value), then 32 elements per round
        p.f16acc = true;
        p.reduce = 2;
        const size_t residual = dim % 32;
        size_t pos = 0;
        if (residual) {
            int s = new_step();
            for (size_t j = 0; j < residual; j++) put(s, (int)j, j);
            pos = residual;
        }
        for (; pos < dim; pos += 32) {
            int s = new_step();
            for (size_t j = 0; j < 32; j++) put(s, (int)j, pos + j);
        }
        return p;
    }
    if (type == VSGPU_F16 && dim < 16) {
        // F16C tier at dims 8..15 (L2_F16C_FP16.h:28-83, IP_F16C_FP16.h:27-81): the first dim-8 elements go through a
        // zero blend into sum0 (lanes 0..7), the remaining 8-block into sum1 (lanes 8..15); sum2 and sum3 stay zero.
        // A multiply and an fma onto a zero accumulator round alike, so one fused step covers both heads.
        p.reduce = 1;
        const size_t r8 = dim - 8;
        int s = new_step();
        for (size_t j = 0; j < r8; j++) put(s, (int)j, j);
        for (size_t j = 0; j < 8; j++) put(s, (int)(8 + j), r8 + j);
        return p;
    }

    if (type == VSGPU_F32 || type == VSGPU_F16 || type == VSGPU_F64 || type == VSGPU_SQ8) {
        // two accumulators of h lanes each (h = 16 for 32-bit math, 8 for fp64):
        // L2_AVX512F_FP32.h:21-59, L2_AVX512F_FP16.h, L2_AVX512F_FP64.h:21-59 and the IP twins; the SQ8 x FP32
        // kernel (IP_AVX512F_BW_VL_VNNI_SQ8_FP32.h:49-104) has the same shape over float(code) * y.
        const size_t h = vl / 2, chunk = vl;
        const size_t residual = dim % chunk, rh = residual % h;
        size_t pos = 0;
        if (residual) {
            int s = new_step();
            for (size_t j = 0; j < rh; j++) put(s, (int)j, j);  // masked head -> sum0
            pos = rh;
            if (residual >= h) {  // one full h-block -> sum1
                for (size_t j = 0; j < h; j++) put(s, (int)(h + j), pos + j);
                pos += h;
            }
        }
        for (; pos < dim; pos += chunk) {
            int s = new_step();
            for (size_t j = 0; j < chunk; j++) put(s, (int)j, pos + j);  // lanes 0..h-1 sum0, rest sum1
        }
        return p;
    }

    // bf16, 16 fp32 lanes, one accumulator
    const size_t residual = dim % 32;
    size_t pos = 0;
    if ((tier == VSGPU_TIER_AVX512_BF16 || tier == VSGPU_TIER_AVX512_FP16) && !p.is_l2) {   // (every avx512_fp16 CPU has avx512_bf16)
        // vdpbf16ps (IP_AVX512_BF16_VL_BF16.h:14-47): lane j takes the pair (2j, 2j+1), the odd
        // element first, each accumulated with its own rounding (characterised on hardware, see
        // oracle/vso.c).
        p.dpbf16 = true;
        if (residual) {
            int s1 = new_step(), s0 = new_step();
            for (int j = 0; j < 16; j++) {
                if ((size_t)(2 * j + 1) < residual) put(s1, j, 2 * j + 1);
                if ((size_t)(2 * j) < residual) put(s0, j, 2 * j);
            }
            pos = residual;
        }
        for (; pos < dim; pos += 32) {
            int s1 = new_step(), s0 = new_step();
            for (int j = 0; j < 16; j++) {
                put(s1, j, pos + 2 * j + 1);
                put(s0, j, pos + 2 * j);
            }
        }
        return p;
    }
    // VBMI2 tier (L2_AVX512BW_VBMI2_BF16.h:40-78, IP_AVX512BW_VBMI2_BF16.h:38-76)
    if (residual) {
        if (residual >= 16) {
            int s = new_step();
            for (int j = 0; j < 16; j++) put(s, j, j);
            pos = 16;
        }
        if (residual != 16) {
            int s = new_step();
            size_t r = residual % 16;
            for (size_t j = 0; j < r; j++) put(s, (int)j, pos + j);
            pos += r;
        }
    }
    for (; pos < dim; pos += 32) {
        int lo = new_step(), hi = new_step();
        for (int j = 0; j < 16; j++) {
            int L = j / 4, w = j % 4;
            put(lo, j, pos + 8 * L + w);      // unpacklo: elements 0..3 of each 128-bit lane
            put(hi, j, pos + 8 * L + 4 + w);  // unpackhi: elements 4..7
        }
    }
    return p;
}

}  // namespace vsg
