// vsgpu_lowp.hip -- bf16 / fp16 / int8 / uint8 MFMA filter path of vsgpu_topk (kernels: mfma_lowp_kernels.hpp)
#include "vsgpu_internal.hpp"
#include "mfma_lowp_kernels.hpp"
#include "mfma_i8x32_kernels.hpp"
#include "mfma_wide_kernels.hpp"
// (the diagnosis build of the filter kernels -- run-time dbg switches -- covers narrower widths in a tuning build)
#ifdef VSGPU_TUNING
constexpr int LOWP_DIAG_MIN_KS = 12;
#else
constexpr int LOWP_DIAG_MIN_KS = 16;
#endif

using namespace vsg;

// ------------------------------------------------------------------ low-precision MFMA filter path
// One launcher for every instantiation: ring depths above 3 slots need more than the default 64 KiB of
// dynamic LDS, which HIP only grants after the attribute is raised.
template <int LK, int KS, int MODE, int RT, int NW, int NQW, int MINW, int NS, int STAGE = MF_STAGE_BYTES, int DIST = 0, int DLATE = 0, int ISS = 0>
static void launch_lowp_k(const LowpParams &P, dim3 grid, hipStream_t s) {
    constexpr int lds_bytes = lowp_lds_bytes(NW, KS, RT, NS, STAGE, false, MODE == MF_PROBE ? NQW : 0);
    static_assert(lds_bytes <= 160 * 1024, "LDS ring does not fit");
    auto go = [&](auto kern) {
        if (lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds_bytes, s, P);
    };
    // the diagnosis build (run-time dbg switches, paired query tiles) exists for the plain 3-slot filter kernels only
    constexpr bool has_diag = MODE == MF_FILTER && NS == 3 && DIST == 0 && DLATE == 0 && KS >= LOWP_DIAG_MIN_KS && ISS == 0;
    if constexpr (has_diag) {
        if (P.dbg || P.pair_map) return go(k_mfma_filter_lowp<LK, KS, MODE, RT, NW, NQW, MINW, NS, STAGE, false, DIST, DLATE, true>);
    }
    // (no diagnosis build of this variant: the switches are compiled out, the production kernel runs)
    go(k_mfma_filter_lowp<LK, KS, MODE, RT, NW, NQW, MINW, NS, STAGE, false, DIST, DLATE, false, ISS>);
}
template <int LK, int KS, int RT, int NQW>
static void launch_lowp_t(int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    if (mode == MF_PROBE) launch_lowp_k<LK, KS, MF_PROBE, RT, 8, NQW, 1, 3>(P, grid, s);
    else launch_lowp_k<LK, KS, MF_FILTER, RT, 8, NQW, 1, 3>(P, grid, s);
}
// Batches of at most 64 queries: 4 waves x 16 queries per workgroup (two or more resident per CU, each streaming its own row
// tiles, like the fp32 filter) instead of an 8-wave workgroup whose upper half multiplies and screens padding
template <int LK, int KS, int RT>
static void launch_lowp_narrow(int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    if (mode == MF_PROBE) launch_lowp_k<LK, KS, MF_PROBE, RT, 4, 1, 2, 3>(P, grid, s);
    else launch_lowp_k<LK, KS, MF_FILTER, RT, 4, 1, 2, 3>(P, grid, s);
}
// returns false when the table's shape has no 4-wave instance
static bool launch_lowp_narrow_any(const vsgpu_table *t, int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    if (t->lp_kind == LP_SQ8) {
        switch (t->lp_ksteps) {
        case 8: launch_lowp_narrow<LP_SQ8, 8, 64>(mode, P, grid, s); return true;
        case 12: launch_lowp_narrow<LP_SQ8, 12, 64>(mode, P, grid, s); return true;
        default: launch_lowp_narrow<LP_SQ8, 16, 64>(mode, P, grid, s); return true;
        }
    }
    if (t->lp_kind == LP_BF16 || t->lp_kind == LP_F16) {
        const bool bf = t->lp_kind == LP_BF16;
        switch (t->lp_ksteps) {
        case 8: bf ? launch_lowp_narrow<LP_BF16, 8, 64>(mode, P, grid, s) : launch_lowp_narrow<LP_F16, 8, 64>(mode, P, grid, s); return true;
        case 16: bf ? launch_lowp_narrow<LP_BF16, 16, 32>(mode, P, grid, s) : launch_lowp_narrow<LP_F16, 16, 32>(mode, P, grid, s); return true;
        case 24: bf ? launch_lowp_narrow<LP_BF16, 24, 32>(mode, P, grid, s) : launch_lowp_narrow<LP_F16, 24, 32>(mode, P, grid, s); return true;
        case 32: bf ? launch_lowp_narrow<LP_BF16, 32, 16>(mode, P, grid, s) : launch_lowp_narrow<LP_F16, 32, 16>(mode, P, grid, s); return true;   // (round 5: 1024 elements)
        default: return false;
        }
    }
    if ((t->lp_kind == LP_I8 || t->lp_kind == LP_U8 || t->lp_kind == LP_U8C) && (t->lp_ksteps == 8 || t->lp_ksteps == 12) && t->lp_rt == 64) {
        // int8 / uint8 rows of at most 768 elements, at most 128 queries (round 5): the 16-wave, 256-query workgroup these widths had
        // -- one per CU, half of it padding at 128 queries -- streamed 1.6-2.0 TB/s at every batch size (profiles/r05_anomaly_scan.txt);
        // 8 waves x 16 queries at <= 128 VGPRs, two workgroups per CU, four ring slots: the SQ8 filter's shape
        auto go = [&](auto ks_tag, auto lk_tag) {
            constexpr int KS = decltype(ks_tag)::value, LK = decltype(lk_tag)::value;
            // (registers: at 12 k-steps the int8 / uint8 L2 / IP kernel needs more than the 128 that two workgroups per CU leave a wave
            // -- 42-54 spilled -- and runs better alone on its CU with 256: int8 L2 768 3.2 -> 4.35 TB/s, Cosine 4.6 -> 4.45; the 8 k-step
            // kernels and uint8 Cosine fit and lose 12-19 % alone: profiles/r05_anomaly_fixes.txt)
            constexpr int MW = (KS == 12 && LK != LP_U8C) ? 2 : 4;
            if (mode == MF_PROBE) launch_lowp_k<LK, KS, MF_PROBE, 64, 8, 1, MW, 3>(P, grid, s);
            else launch_lowp_k<LK, KS, MF_FILTER, 64, 8, 1, MW, 4>(P, grid, s);
        };
        auto by_kind = [&](auto ks_tag) {
            if (t->lp_kind == LP_I8) go(ks_tag, std::integral_constant<int, LP_I8>{});
            else if (t->lp_kind == LP_U8) go(ks_tag, std::integral_constant<int, LP_U8>{});
            else go(ks_tag, std::integral_constant<int, LP_U8C>{});
        };
        if (t->lp_ksteps == 8) by_kind(std::integral_constant<int, 8>{});
        else by_kind(std::integral_constant<int, 12>{});
        return true;
    }
    if ((t->lp_kind == LP_I8 || t->lp_kind == LP_U8 || t->lp_kind == LP_U8C) && t->lp_ksteps == 16 && t->lp_rt == 32) {
        // int8 / uint8, width 1024, at most 128 queries: 8 waves x 16 queries at <= 128 VGPRs, two workgroups resident per CU
        // (uint8 Cosine joined in round 5: on the 16-wave workgroup it ran 2.1 TB/s between 3.8 at 768 and 3.9 at 2048 elements)
        if (t->lp_kind == LP_I8) {
            if (mode == MF_PROBE) launch_lowp_k<LP_I8, 16, MF_PROBE, 32, 8, 1, 4, 3>(P, grid, s);
            else launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 8, 1, 4, 3>(P, grid, s);
        } else if (t->lp_kind == LP_U8C) {
            if (mode == MF_PROBE) launch_lowp_k<LP_U8C, 16, MF_PROBE, 32, 8, 1, 4, 3>(P, grid, s);
            else launch_lowp_k<LP_U8C, 16, MF_FILTER, 32, 8, 1, 4, 3>(P, grid, s);
        } else {
            if (mode == MF_PROBE) launch_lowp_k<LP_U8, 16, MF_PROBE, 32, 8, 1, 4, 3>(P, grid, s);
            else launch_lowp_k<LP_U8, 16, MF_FILTER, 32, 8, 1, 4, 3>(P, grid, s);
        }
        return true;
    }
    return false;
}
// query-tile width of the narrow-batch kernels (0: none for this table)
static size_t lowp_narrow_qtile(const vsgpu_table *t) {
    if (t->lp_kind == LP_SQ8) return 64;
    if ((t->lp_kind == LP_BF16 || t->lp_kind == LP_F16) && t->lp_ksteps <= 32) return 64;
    if ((t->lp_kind == LP_I8 || t->lp_kind == LP_U8 || t->lp_kind == LP_U8C) && t->lp_ksteps == 16 && t->lp_rt == 32) return 128;
    // int8 / uint8 (every metric) up to 768 elements: the same 8-wave shape the SQ8 filter runs on (rows of 768 + 16 bytes there)
    if ((t->lp_kind == LP_I8 || t->lp_kind == LP_U8 || t->lp_kind == LP_U8C) && (t->lp_ksteps == 8 || t->lp_ksteps == 12) && t->lp_rt == 64) return 128;
    return 0;
}
template <int KS, int RT, int LK = LP_I8> static void launch_lowp_i8(int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    if (mode == MF_PROBE) launch_lowp_k<LK, KS, MF_PROBE, RT, 16, 1, 1, 3>(P, grid, s);
    else launch_lowp_k<LK, KS, MF_FILTER, RT, 16, 1, 1, 3>(P, grid, s);
}
template <int LK> static void launch_lowp_h16(int ks, int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    switch (ks) {
    case 8: launch_lowp_t<LK, 8, 64, 1>(mode, P, grid, s); break;
    case 16: launch_lowp_t<LK, 16, 32, 1>(mode, P, grid, s); break;
    case 24: launch_lowp_t<LK, 24, 32, 1>(mode, P, grid, s); break;
    case 48: launch_lowp_t<LK, 48, 16, 1>(mode, P, grid, s); break;  // d = 1536: 192 VGPRs of query fragments per wave
    case 64:   // d = 2048: 4 waves x 16 queries, 256 registers of fragments per wave (AGPRs)
        if (mode == MF_PROBE) launch_lowp_k<LK, 64, MF_PROBE, 16, 4, 1, 1, 3>(P, grid, s);
        else launch_lowp_k<LK, 64, MF_FILTER, 16, 4, 1, 1, 3>(P, grid, s);
        break;
    default: launch_lowp_t<LK, 32, 16, 1>(mode, P, grid, s); break;
    }
}
// Measured-slower variants of these kernels (barrier-free ring, K-split int8 kernel, query-split bf16 tiles, ring-depth sweeps)
// live in tools/tuning_kernels/ and are compiled only by `make TUNING=1`: lowp_tuning.inc defines the hooks, the shipped build
// has these stubs.
#ifdef VSGPU_TUNING
#include "lowp_tuning.inc"
#else
static bool launch_lowp_variant(const vsgpu_table *, int, LowpParams, uint32_t, unsigned, hipStream_t) { return false; }
static bool tuning_launch_lowp(const vsgpu_table *, int, const LowpParams &, dim3, hipStream_t) { return false; }
static bool tuning_ksplit_on(const vsgpu_ctx *) { return false; }
static bool tuning_hsplit(const vsgpu_table *, const vsgpu_ctx *, size_t) { return false; }
template <int LK> static void launch_lowp_h16_split(int, const LowpParams &, dim3, hipStream_t) {}
#endif
// int8 / uint8 rows of 1025 .. 2048 elements: 8 waves x 16 queries, 16-row tiles of two 16 KiB stages
template <int LK> static void launch_lowp_w2048(int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    if (mode == MF_PROBE) launch_lowp_k<LK, 32, MF_PROBE, 16, 8, 1, 1, 3>(P, grid, s);
    else launch_lowp_k<LK, 32, MF_FILTER, 16, 8, 1, 1, 3>(P, grid, s);
}
template <int LK> static void launch_lowp_w3072(int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    if constexpr (LK == LP_U8C) {   // 4 waves x 16 queries, one wave per SIMD (vsgpu.hip: lp_qtile 64)
        if (mode == MF_PROBE) launch_lowp_k<LK, 48, MF_PROBE, 16, 4, 1, 1, 3>(P, grid, s);
        else launch_lowp_k<LK, 48, MF_FILTER, 16, 4, 1, 1, 3>(P, grid, s);
        return;
    }
    if (mode == MF_PROBE) launch_lowp_k<LK, 48, MF_PROBE, 16, 8, 1, 1, 3>(P, grid, s);
    else launch_lowp_k<LK, 48, MF_FILTER, 16, 8, 1, 1, 3>(P, grid, s);
}
template <int LK> static void launch_lowp_w4096(int mode, const LowpParams &P, dim3 grid, hipStream_t s) {   // 4 waves, fragments in AGPRs
    if (mode == MF_PROBE) launch_lowp_k<LK, 64, MF_PROBE, 16, 4, 1, 1, 3>(P, grid, s);
    else launch_lowp_k<LK, 64, MF_FILTER, 16, 4, 1, 1, 3>(P, grid, s);
}
static void launch_lowp(const vsgpu_table *t, int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    if (tuning_launch_lowp(t, mode, P, grid, s)) return;   // (tuning build only)
    if (t->lp_kind == LP_BF16) launch_lowp_h16<LP_BF16>(t->lp_ksteps, mode, P, grid, s);
    else if (t->lp_kind == LP_F16) launch_lowp_h16<LP_F16>(t->lp_ksteps, mode, P, grid, s);
    else if (t->lp_kind == LP_U8C) {
        switch (t->lp_ksteps) {
        case 24: launch_lowp_t<LP_U8C, 24, 32, 1>(mode, P, grid, s); break;   // width 1536
        case 32: launch_lowp_w2048<LP_U8C>(mode, P, grid, s); break;
        case 48: launch_lowp_w3072<LP_U8C>(mode, P, grid, s); break;
        case 64: launch_lowp_w4096<LP_U8C>(mode, P, grid, s); break;
        case 8: launch_lowp_i8<8, 64, LP_U8C>(mode, P, grid, s); break;
        case 12: launch_lowp_i8<12, 64, LP_U8C>(mode, P, grid, s); break;
        default:
            if (mode == MF_FILTER) launch_lowp_k<LP_U8C, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768>(P, grid, s);
            else launch_lowp_k<LP_U8C, 16, MF_PROBE, 32, 16, 1, 1, 3, 32768>(P, grid, s);   // the filter's whole-row slots
            break;
        }
    } else if (t->lp_kind == LP_SQ8) {
        // filter, widths 512 / 768: 128 registers and a 4-slot ring, two workgroups per CU -- one's epilogue and refill requests
        // run under the other's MFMA stream (10 M x 768 L2, batch 128: 1.83 ms at one workgroup per CU, 1.68 with two, 1.63 with 4 slots)
        switch (t->lp_ksteps) {
        case 8:
            if (mode == MF_FILTER) launch_lowp_k<LP_SQ8, 8, MF_FILTER, 64, 8, 1, 4, 4>(P, grid, s);
            else launch_lowp_t<LP_SQ8, 8, 64, 1>(mode, P, grid, s);
            break;
        case 12:
            if (mode == MF_FILTER) launch_lowp_k<LP_SQ8, 12, MF_FILTER, 64, 8, 1, 4, 4>(P, grid, s);
            else launch_lowp_t<LP_SQ8, 12, 64, 1>(mode, P, grid, s);
            break;
        default: launch_lowp_t<LP_SQ8, 16, 64, 1>(mode, P, grid, s); break;
        }
    } else if (t->lp_kind == LP_U8) {
        switch (t->lp_ksteps) {
        case 24: launch_lowp_t<LP_U8, 24, 32, 1>(mode, P, grid, s); break;   // width 1536
        case 32: launch_lowp_w2048<LP_U8>(mode, P, grid, s); break;
        case 48: launch_lowp_w3072<LP_U8>(mode, P, grid, s); break;
        case 64: launch_lowp_w4096<LP_U8>(mode, P, grid, s); break;
        case 8: launch_lowp_i8<8, 64, LP_U8>(mode, P, grid, s); break;
        case 12: launch_lowp_i8<12, 64, LP_U8>(mode, P, grid, s); break;
        default:
            if (mode == MF_FILTER) launch_lowp_k<LP_U8, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768>(P, grid, s);
            else launch_lowp_k<LP_U8, 16, MF_PROBE, 32, 16, 1, 1, 3, 32768>(P, grid, s);   // the filter's whole-row slots
            break;
        }
    } else {
        // 16 waves x 16 queries.  d=1024 filter: a ring slot holds 32 whole rows (1 KiB per DMA instruction, one
        // barrier per 32 KiB): 3.54 TB/s against 3.24 for 16 KiB half-row slots, 3.1 for 8 waves x 32 queries and
        // 2.5 for 4 waves x 64 queries (profiles/r01_tuning_lowp.txt)
        switch (t->lp_ksteps) {
        case 24: launch_lowp_t<LP_I8, 24, 32, 1>(mode, P, grid, s); break;   // width 1536
        case 32: launch_lowp_w2048<LP_I8>(mode, P, grid, s); break;
        case 48: launch_lowp_w3072<LP_I8>(mode, P, grid, s); break;
        case 64: launch_lowp_w4096<LP_I8>(mode, P, grid, s); break;
        case 8: launch_lowp_i8<8, 64>(mode, P, grid, s); break;
        case 12: launch_lowp_i8<12, 64>(mode, P, grid, s); break;
        default:
            if (mode == MF_FILTER) launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 16, 1, 1, 3, 32768>(P, grid, s);
            else launch_lowp_k<LP_I8, 16, MF_PROBE, 32, 16, 1, 1, 3, 32768>(P, grid, s);   // the filter's whole-row slots
            break;
        }
    }
}

// 32x32x32 filter for 1 KiB int8 / uint8 rows (mfma_i8x32_kernels.hpp); var = VAR bits of the kernel (tuning build; the shipped
// build runs VAR = 32769: requests by waves 0-3, fragments prefetched across the barrier)
template <int LK, int EPI> static void launch_i8_x32_e(int var, const LowpParams &P, dim3 grid, hipStream_t s) {
    auto go = [&](auto kern, int ns) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, x32_lds_bytes(ns));
        hipLaunchKernelGGL(kern, grid, dim3(X32_NW * 64), x32_lds_bytes(ns), s, P);
    };
    switch (var) {
#ifdef VSGPU_TUNING
#define X32_CASE(V) case V: go(k_i8_filter_x32<LK, EPI, V>, 4); break;
    X32_CASE(0) X32_CASE(1) X32_CASE(2) X32_CASE(3) X32_CASE(8) X32_CASE(16) X32_CASE(32768) X32_CASE(32768 + 2) X32_CASE(32769 + 2)
    X32_CASE(16384) X32_CASE(16385) X32_CASE(32768 + 16384) X32_CASE(32769 + 16384)
    X32_CASE(32) X32_CASE(64) X32_CASE(128) X32_CASE(256) X32_CASE(512) X32_CASE(32769 + 128) X32_CASE(32769 + 256) X32_CASE(32769 + 512) X32_CASE(32769 + 32)
    case 65536 + 32769: go(k_i8_filter_x32<LK, EPI, 32769, 3, 2>, 3); break;
    case 131072 + 32769: go(k_i8_filter_x32<LK, EPI, 32769, 4, 2>, 4); break;            // the shipped stream, two units ahead
#define X32_SHIFT_CASE(V) case V: go(k_i8_filter_x32<LK, EPI, V, 4, 2>, 4); break;      // SHIFT variants: 4 slots, 2 ahead
    X32_SHIFT_CASE(32769 + 4) X32_SHIFT_CASE(32769 + 4 + 16384) X32_SHIFT_CASE(32769 + 4 + 16) X32_SHIFT_CASE(32769 + 4 + 512)
    X32_SHIFT_CASE(32769 + 4 + 128) X32_SHIFT_CASE(32769 + 4 + 32) X32_SHIFT_CASE(32769 + 4 + 2) X32_SHIFT_CASE(32769 + 4 + 16384 + 512)
    X32_SHIFT_CASE(32768 + 4) X32_SHIFT_CASE(32769 + 4 + 8)
    // FREE: no barrier in the loop (two LDS counters), 4 slots, 2 units ahead
    X32_SHIFT_CASE(32769 + 1024) X32_SHIFT_CASE(32769 + 1024 + 16384) X32_SHIFT_CASE(32769 + 1024 + 512) X32_SHIFT_CASE(32769 + 1024 + 128)
    X32_SHIFT_CASE(32769 + 1024 + 32) X32_SHIFT_CASE(32769 + 1024 + 16) X32_SHIFT_CASE(32769 + 1024 + 8)
#undef X32_SHIFT_CASE
    X32_CASE(32769 + 16384 + 512) X32_CASE(32769 + 16384 + 128)
#undef X32_CASE
#endif
    default: go(k_i8_filter_x32<LK, EPI, 32769>, 4); break;
    }
}
// k_i8_filter_x32l (the lean stream): var bit 262144 selects it, bit 4 = SHIFT, bit 131072 = ring 4 / 2 without SHIFT
template <int LK, int EPI> static void launch_i8_x32l_e(int var, const LowpParams &P, dim3 grid, hipStream_t s) {
    auto go = [&](auto kern, int ns) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, x32_lds_bytes(ns));
        hipLaunchKernelGGL(kern, grid, dim3(X32_NW * 64), x32_lds_bytes(ns), s, P);
    };
    // shipped: SHIFT (waves 4-7 meet the barrier mid-stream) + counted fragment waits for int8 rows; the uint8 stream's sign flips
    // leave no registers for the inline-asm reads (they would spill): SHIFT alone
    constexpr int SHIP = LK == LP_U8 ? 5 : 5 + 16384;
#ifdef VSGPU_TUNING
    if ((var & 4) && (var & 16384) && (var & 8)) go(k_i8_filter_x32l<LK, EPI, 5 + 16384 + 8, 4, 2>, 4);
    else if ((var & 4) && (var & 16384) && (var & 32)) go(k_i8_filter_x32l<LK, EPI, 5 + 16384 + 32, 4, 2>, 4);
    else if ((var & 4) && (var & 16384) && (var & 256) && (var & 1024)) go(k_i8_filter_x32l<LK, EPI, 5 + 16384 + 256 + 1024, 4, 2>, 4);
    else if ((var & 4) && (var & 16384) && (var & 1024)) go(k_i8_filter_x32l<LK, EPI, 5 + 16384 + 1024, 4, 2>, 4);
    else if ((var & 4) && (var & 16384) && (var & 128) && (var & 512)) go(k_i8_filter_x32l<LK, EPI, 5 + 16384 + 128 + 512, 4, 2>, 4);
    else if ((var & 4) && (var & 16384) && (var & 128) && (var & 64)) go(k_i8_filter_x32l<LK, EPI, 5 + 16384 + 128 + 64, 4, 2>, 4);
    else if ((var & 4) && (var & 16384) && (var & 128)) go(k_i8_filter_x32l<LK, EPI, 5 + 16384 + 128, 4, 2>, 4);
    else if ((var & 4) && (var & 16384) && (var & 256)) go(k_i8_filter_x32l<LK, EPI, 5 + 16384 + 256, 4, 2>, 4);   // per-unit thresholds: no gain at 50 M rows
    else if ((var & 4) && (var & 16384)) go(k_i8_filter_x32l<LK, EPI, SHIP, 4, 2>, 4);
    else if ((var & 16384) && (var & 8)) go(k_i8_filter_x32l<LK, EPI, 1 + 16384 + 8, 4, 3>, 4);
    else if ((var & 4) && (var & 8)) go(k_i8_filter_x32l<LK, EPI, 5 + 8, 4, 2>, 4);
    else if (var & 4) go(k_i8_filter_x32l<LK, EPI, 5, 4, 2>, 4);
    else if (var & 131072) go(k_i8_filter_x32l<LK, EPI, 1, 4, 2>, 4);
    else if (var & 16384) go(k_i8_filter_x32l<LK, EPI, 1 + 16384, 4, 3>, 4);
    else go(k_i8_filter_x32l<LK, EPI, 1, 4, 3>, 4);
#else
    (void)var;
    go(k_i8_filter_x32l<LK, EPI, SHIP, 4, 2>, 4);
#endif
}
static void launch_i8_x32(const vsgpu_table *t, int var, const LowpParams &P, dim3 grid, hipStream_t s) {
    if (var & 262144) {
        if (t->lp_kind == LP_U8) {
            var &= ~16384;   // (see SHIP above)
            if (P.epi == LE_U8_IP) launch_i8_x32l_e<LP_U8, LE_U8_IP>(var, P, grid, s);
            else launch_i8_x32l_e<LP_U8, LE_I8_L2>(var, P, grid, s);
        } else if (P.epi == LE_I8_COS) launch_i8_x32l_e<LP_I8, LE_I8_COS>(var, P, grid, s);
        else if (P.epi == LE_I8_L2) launch_i8_x32l_e<LP_I8, LE_I8_L2>(var, P, grid, s);
        else launch_i8_x32l_e<LP_I8, LE_I8_IP>(var, P, grid, s);
        return;
    }
    if (t->lp_kind == LP_U8) {
        if (P.epi == LE_U8_IP) launch_i8_x32_e<LP_U8, LE_U8_IP>(var, P, grid, s);
        else launch_i8_x32_e<LP_U8, LE_I8_L2>(var, P, grid, s);
    } else if (P.epi == LE_I8_COS) launch_i8_x32_e<LP_I8, LE_I8_COS>(var, P, grid, s);
    else if (P.epi == LE_I8_L2) launch_i8_x32_e<LP_I8, LE_I8_L2>(var, P, grid, s);
    else launch_i8_x32_e<LP_I8, LE_I8_IP>(var, P, grid, s);
}

// int8 with the query batch split over two 8-wave workgroups (blockIdx.y): both stream the same row tiles, the
// second reader is expected to hit L2 (same XCD when gridDim.x % 8 == 0)
static void launch_lowp_i8_split(const vsgpu_table *t, int mode, const LowpParams &P, dim3 grid, hipStream_t s) {
    if (t->lp_ksteps == 16 && t->lp_rt == 32) {
        if (mode == MF_PROBE) launch_lowp_k<LP_I8, 16, MF_PROBE, 32, 8, 1, 2, 3>(P, grid, s);
        else if (grid.y == 2 && grid.x % 8 == 0) {
            // both query tiles resident together: 2 x 8 waves per CU need <= 128 VGPRs (4 waves per SIMD)
            LowpParams Q = P;
            Q.pair_map = 1;
            launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 8, 1, 4, 3>(Q, dim3(grid.x * 2), s);
        } else launch_lowp_k<LP_I8, 16, MF_FILTER, 32, 8, 1, 2, 3>(P, grid, s);
    } else if (t->lp_ksteps == 12) {
        if (mode == MF_PROBE) launch_lowp_k<LP_I8, 12, MF_PROBE, 64, 8, 1, 2, 3>(P, grid, s);
        else launch_lowp_k<LP_I8, 12, MF_FILTER, 64, 8, 1, 2, 3>(P, grid, s);
    } else {
        if (mode == MF_PROBE) launch_lowp_k<LP_I8, 8, MF_PROBE, 64, 8, 1, 2, 3>(P, grid, s);
        else launch_lowp_k<LP_I8, 8, MF_FILTER, 64, 8, 1, 2, 3>(P, grid, s);
    }
}


// bf16 / fp16 rows of 2049 .. 8192 elements on k_mfma_filter_wide (mfma_wide_kernels.hpp): same records, same bound
#ifndef WIDE_NS_ALONE
#define WIDE_NS_ALONE 5
#endif
#ifndef WIDE_NW4
#define WIDE_NW4 4   // waves of the four-column-block workgroup (8: measured at half the rate, see k_mfma_filter_wide)
#endif
template <int EK, int MODE> static void launch_wide_h16_m(int ksteps, int nq_blocks, const MfmaParams &P, dim3 grid, hipStream_t s) {
    auto go = [&](auto kern, int nqb, int ns, int nw = 4) {
        const int lds_bytes = mfw_lds_bytes(MODE == MF_PROBE, nqb, ns, EK, nw);
        if (lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        hipLaunchKernelGGL(kern, grid, dim3(64 * nw), lds_bytes, s, P);
    };
    // two 16-query column blocks per workgroup up to width 6144 (the fragments of 32 queries fit the registers of a wave);
    // WIDE_NS_ALONE: see vsgpu_mfma.hip and k_mfma_filter_wide (ring depth of a workgroup alone on its CU)
    constexpr int NA = WIDE_NS_ALONE;
    switch (ksteps) {
    case 96:   // (four column blocks -- 64 queries in ONE pass over the rows -- fit a wave's 512 registers at this width: half of them AGPRs)
        if (nq_blocks == 4) go(k_mfma_filter_wide<96, MODE, 0, EK, 4, NA, WIDE_NW4>, 4, NA, WIDE_NW4);
        else nq_blocks == 2 ? go(k_mfma_filter_wide<96, MODE, 0, EK, 2, NA>, 2, NA) : go(k_mfma_filter_wide<96, MODE, 0, EK, 1, 3>, 1, 3);
        break;
    case 128:   // (three column blocks are the same 384 registers of fragments as four at width 96)
        if (nq_blocks == 3) go(k_mfma_filter_wide<128, MODE, 0, EK, 3, NA>, 3, NA);
        else nq_blocks == 2 ? go(k_mfma_filter_wide<128, MODE, 0, EK, 2, NA>, 2, NA) : go(k_mfma_filter_wide<128, MODE, 0, EK, 1, 3>, 1, 3);
        break;
    case 192: nq_blocks == 2 ? go(k_mfma_filter_wide<192, MODE, 0, EK, 2, NA>, 2, NA) : go(k_mfma_filter_wide<192, MODE, 0, EK, 1, 3>, 1, 3); break;
    default: go(k_mfma_filter_wide<256, MODE, 0, EK, 1, NA>, 1, NA); break;
    }
}
static void launch_wide_h16(const vsgpu_table *t, int mode, int nq_blocks, const LowpParams &L, dim3 grid, hipStream_t s) {
    MfmaParams P{};
    P.slabs = L.slabs;
    P.norm_slabs = reinterpret_cast<const float *const *>(L.aux_slabs);
    P.slab_shift = L.slab_shift;
    P.slab_mask = L.slab_mask;
    P.row_stride = L.row_stride;
    P.n_rows = L.n_rows;
    P.tile_first = L.tile_first;
    P.tile_step = L.tile_step;
    P.n_tiles = L.n_tiles;
    P.tile_run_shift = L.tile_run_shift;
    P.qfrag = L.qfrag;
    P.qn2 = reinterpret_cast<const float *>(L.qaux);
    P.cE = L.cE;
    P.absE = L.absE;
    P.is_l2 = L.epi == LE_FP_L2 ? 1 : 0;
    P.tilemin = L.tilemin;
    P.tilemin_stride = L.tilemin_stride;
    P.tau = L.tau;
    P.counts = L.counts;
    P.cand = L.cand;
    P.cap = L.cap;
    P.iepi = L.epi;
    P.qmeta = L.qmeta;
    if (t->lp_kind == LP_U8C) {   // uint8 Cosine rows of 4097 .. 16384 elements: 16-byte aux records {norm, sum (x - 128)}
        if (mode == MF_PROBE) launch_wide_h16_m<5, MF_PROBE>(t->lp_ksteps, nq_blocks, P, grid, s);
        else launch_wide_h16_m<5, MF_FILTER>(t->lp_ksteps, nq_blocks, P, grid, s);
    } else if (t->lp_kind == LP_I8) {   // int8 / uint8 rows of 4097 .. 16384 elements: exact integer scores, no re-rank
        if (mode == MF_PROBE) launch_wide_h16_m<3, MF_PROBE>(t->lp_ksteps, nq_blocks, P, grid, s);
        else launch_wide_h16_m<3, MF_FILTER>(t->lp_ksteps, nq_blocks, P, grid, s);
    } else if (t->lp_kind == LP_U8) {
        if (mode == MF_PROBE) launch_wide_h16_m<4, MF_PROBE>(t->lp_ksteps, nq_blocks, P, grid, s);
        else launch_wide_h16_m<4, MF_FILTER>(t->lp_ksteps, nq_blocks, P, grid, s);
    } else if (t->lp_kind == LP_BF16) {
        if (mode == MF_PROBE) launch_wide_h16_m<1, MF_PROBE>(t->lp_ksteps, nq_blocks, P, grid, s);
        else launch_wide_h16_m<1, MF_FILTER>(t->lp_ksteps, nq_blocks, P, grid, s);
    } else {
        if (mode == MF_PROBE) launch_wide_h16_m<2, MF_PROBE>(t->lp_ksteps, nq_blocks, P, grid, s);
        else launch_wide_h16_m<2, MF_FILTER>(t->lp_ksteps, nq_blocks, P, grid, s);
    }
}

int topk_lowp(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k, size_t cap,
                     uint32_t *ids, double *scores, uint32_t *counts) {
    vsgpu_ctx *c = t->ctx;
    const size_t n = t->n, dim = t->dim;
    const int KS = t->lp_ksteps, RT = t->lp_rt;
    const bool qsplit = t->lp_kind == LP_I8 && c->opt_lowp_qsplit;
    const bool hsplit = tuning_hsplit(t, c, nq);   // (tuning build only: 64-query bf16 tiles, measured slower)
    const bool narrow = !hsplit && !qsplit && lowp_narrow_qtile(t) && nq <= lowp_narrow_qtile(t) && c->opt_lowp_narrow;
    // int8 / uint8 rows of kernel width 1024, more than 128 queries: the filter pass runs on the 32 x 32 x 32 kernel
    bool x32_lean = false;   // (which of the two 32 x 32 x 32 filters ran: the kernel name in the statistics)
    const bool x32 = c->opt_lowp_x32 && !narrow && !qsplit && (t->lp_kind == LP_I8 || t->lp_kind == LP_U8) && t->lp_ksteps == 16 &&
                     t->lp_rt == 32 && t->lp_qtile == X32_QT && !c->opt_lowp_variant && !c->opt_lowp_dbg
                     && !tuning_ksplit_on(c)
        ;
    // k_mfma_filter_wide: 32 queries per workgroup wherever the registers allow (every width but the widest: 2 x KSTEPS / 4 fragments
    // per wave).  Since every wave works in every unit (round 4) two 32-query tiles beat four 16-query ones at every width
    // (bf16 3072 / 4096, batch 64: 3.04 / 3.19 -> 4.15 / 4.55 TB/s; int8 6144 / 8192: 3.05 / 3.26 -> 3.73 / 4.04); option
    // wide_blocks = 1: 16 everywhere
    // (round 4, late) at kernel width 96 k-steps -- 16-bit rows up to 3072 elements, 8-bit rows up to 6144 -- FOUR column blocks fit: 384
    // registers of fragments per wave, half of them AGPRs, one wave per SIMD; a batch of 64 then crosses the rows once instead of twice
    // (bf16 3072, batch 64: 4.18 -> 6.05 TB/s; int8 6144: 3.74 -> 4.91; profiles/r04_wide_blocks4.txt); option wide_blocks = 2: at most 32
    // (round 5) at kernel width 128 k-steps THREE column blocks fit the same way: used where 48-query tiles cross the rows less often than
    // 32-query tiles do -- 33 .. 48 queries (once instead of twice), 65 .. 96 (2 / 3), 97 .. 128 (3 / 4) ...
    const bool three = t->lp_wide && KS == 128 && (c->opt_wide_blocks == 0 || c->opt_wide_blocks == 3) && (nq + 47) / 48 < (nq + 31) / 32;
    const int wide_blocks = (t->lp_wide && nq > 32 && KS == 96 && (c->opt_wide_blocks == 0 || c->opt_wide_blocks == 4)) ? 4
                            : three ? 3
                            : ((t->lp_wide && nq > 16 && c->opt_wide_blocks != 1 && KS <= 192) ? 2 : 1);
    const size_t QT = t->lp_wide ? (size_t)16 * wide_blocks : hsplit ? 64 : (narrow ? lowp_narrow_qtile(t) : (qsplit ? 128 : (size_t)t->lp_qtile));
    const size_t q_tiles = (nq + QT - 1) / QT, nqp = q_tiles * QT;
    const bool is_sq8 = (t->lp_kind == LP_SQ8);
    const bool is_u8c = (t->lp_kind == LP_U8C);
    const bool is_int = (t->lp_kind == LP_I8 || t->lp_kind == LP_U8 || is_u8c);   // exact integer scores, no re-rank
    const bool is_u8 = (t->lp_kind == LP_U8 || is_u8c);
    const size_t eb = (is_int || is_sq8) ? 1 : 2;
    const size_t kelem = (is_int || is_sq8) ? 64 : 32;        // elements per MFMA k-step
    const size_t per_lane = kelem / 4;            // elements per lane per k-step (16 bytes)

    int rc = VSGPU_OK;   // (the exact-order query images of the re-rank are staged behind the filter launch, under the scan)
    // fragments: [q_tile][wave 8][NQW][KSTEPS][lane 64][16 B]
    const size_t kdim = (size_t)KS * kelem;  // kernel width >= dim
    std::vector<unsigned char> frag(nqp * kdim * eb, 0);
    // x32: [group of 32 queries][32 k-steps of 32 bytes][lane = 32 half + query][16 B] (mfma_i8x32_kernels.hpp)
    std::vector<unsigned char> frag32(x32 ? nqp * kdim : 0, 0);
    std::vector<uint32_t> qaux(nqp, 0);
    std::vector<float> tau0(nqp, -INFINITY);
    std::vector<float> qmeta((is_sq8 || is_u8c) ? nqp * 8 : 0, 0.0f);
    // uint8 Cosine: a padding query (threshold -inf) with norm 0 would score -inf on every row whose re-centred sum is positive -- and
    // -inf <= -inf passes: the tile's padding columns flooded the candidate queue (same replies, a third of the rate; found with the
    // 48-query tiles, present in every padded batch before).  Norm +inf makes their score 1.
    if (is_u8c)
        for (size_t q = nq; q < nqp; q++) qmeta[q * 8] = INFINITY;
    std::vector<signed char> yq(is_sq8 ? dim : 0);
    for (size_t q = 0; q < nq; q++) {
        const unsigned char *src = (const unsigned char *)queries + q * qstride;
        if (is_sq8) {
            // one int8 piece per query element: y_i = s Y_i + e_i, |Y_i| <= 127 (LowpOps<LP_SQ8>): 128 sum e_i and |e|_2 go to
            // the kernel; Wref bounds, per unit of the row's delta, the reference's own fp32 accumulation (dim/32 fused steps
            // per lane + the tree; IP_AVX512F_BW_VL_VNNI_SQ8_FP32.h:49-104): <= 2 (dim/32 + 8) 2^-24 * 255 sum |y_i|
            const size_t qeb = t->type == VSGPU_SQ8H ? 2 : 4;   // fp16 queries are widened first (exactly)
            std::vector<float> ywide;
            if (qeb == 2) {
                ywide.resize(dim);
                for (size_t i = 0; i < dim; i++) {
                    uint16_t h;
                    memcpy(&h, src + 2 * i, 2);
                    ywide[i] = widen_f16(h);
                }
            }
            const float *y = qeb == 2 ? ywide.data() : reinterpret_cast<const float *>(src);
            double ymax = 0, yabs = 0;
            for (size_t i = 0; i < dim; i++) {
                ymax = std::max(ymax, (double)std::fabs(y[i]));
                yabs += std::fabs((double)y[i]);
            }
            float sf = (float)(ymax / 127.0);
            if (!(sf > 0.0f) || !std::isfinite(sf)) sf = 1.0f;
            double se = 0, se2 = 0;
            long sy = 0;
            for (size_t i = 0; i < dim; i++) {
                double r = std::nearbyint((double)y[i] / (double)sf);
                if (!(r >= -127.0)) r = -127.0;   // (also NaN)
                if (r > 127.0) r = 127.0;
                yq[i] = (signed char)r;
                sy += (long)r;
                const double e = (double)y[i] - (double)sf * r;
                se += e;
                se2 += e * e;
            }
            // Wref: the reference's fp32 accumulation + the rounding of `ce` below, per unit of the row's delta
            const float ce = (float)(128.0 * se);
            const double Wref = (2.0 * ((double)dim / 32.0 + 8.0) * std::ldexp(1.0, -24) * 255.0 * yabs +
                                 std::ldexp(1.0, -22) * std::fabs(128.0 * se)) * (1.0 + 1e-6);
            float *qm = &qmeta[q * 8];
            const int K = (int)(128 * sy);
            qm[0] = sf;
            memcpy(&qm[1], &K, 4);
            memcpy(&qm[2], src + qeb * dim, 4);                                 // y_sum
            if (t->metric == VSGPU_L2 || t->sq8_centred) memcpy(&qm[3], src + qeb * dim + 4, 4);  // y_sum_squares | IP: the shift y_mean_ip
            qm[4] = std::nextafter((float)Wref, INFINITY);
            qm[5] = ce;
            qm[6] = std::nextafter((float)(std::sqrt(se2) * (1.0 + 1e-6)), INFINITY);   // |e|_2
            src = reinterpret_cast<const unsigned char *>(yq.data());
        }
        // fragments are stored per group of 16 queries in batch order: the kernels index (q_tile * waves + wave) * NQW + nt,
        // which is q / 16 for every tile shape
        const size_t g16 = q / 16, nn = q % 16;
        for (int s = 0; s < KS; s++)
            for (int kq = 0; kq < 4; kq++) {
                const size_t lane = (size_t)kq * 16 + nn;
                unsigned char *dst = &frag[(((g16 * KS + s) * 64) + lane) * 16];
                const size_t e0 = (kelem * s + per_lane * kq) * eb, have = e0 < dim * eb ? std::min<size_t>(16, dim * eb - e0) : 0;
                if (have) memcpy(dst, src + e0, have);
                if (is_u8)
                    for (size_t b = 0; b < have; b++) dst[b] ^= 0x80;  // q - 128 as int8 (columns past dim stay 0)
            }
        if (x32) {
            const size_t g32 = q / 32, n32 = q % 32;
            for (int s = 0; s < X32_KS; s++)
                for (int hh = 0; hh < 2; hh++) {
                    unsigned char *dst = &frag32[(((g32 * X32_KS + s) * 64) + (size_t)hh * 32 + n32) * 16];
                    const size_t e0 = (size_t)32 * s + 16 * hh, have = e0 < dim ? std::min<size_t>(16, dim - e0) : 0;
                    if (have) memcpy(dst, src + e0, have);
                    if (is_u8)
                        for (size_t b = 0; b < have; b++) dst[b] ^= 0x80;
                }
        }
        if (is_sq8) {
        } else if (is_u8) {
            int s1 = 0, s2 = 0;
            for (size_t i = 0; i < dim; i++) {
                const int v = (int)src[i] - 128;
                s1 += v;
                s2 += v * v;
            }
            const int aux = t->epi == EPI_INT_L2 ? s2 : 128 * s1 + 16384 * (int)dim;
            memcpy(&qaux[q], &aux, 4);
            if (is_u8c) {   // {norm_q, 128 sum q' + 128^2 dim}
                memcpy(&qmeta[q * 8], src + dim, 4);
                memcpy(&qmeta[q * 8 + 1], &aux, 4);
            }
        } else if (is_int) {
            if (t->epi == EPI_INT_COS) memcpy(&qaux[q], src + dim, 4);
            else if (t->epi == EPI_INT_L2) {
                int ss = 0;
                for (size_t i = 0; i < dim; i++) ss += (int)(int8_t)src[i] * (int)(int8_t)src[i];
                memcpy(&qaux[q], &ss, 4);
            }
        } else {
            double ss = 0;
            for (size_t i = 0; i < dim; i++) {
                uint16_t h;
                memcpy(&h, src + 2 * i, 2);
                double v = t->type == VSGPU_BF16 ? (double)widen_bf16(h) : (double)widen_f16(h);
                ss += v * v;
            }
            float f = (float)ss;
            memcpy(&qaux[q], &f, 4);
        }
    }
    const uint32_t total_tiles = (uint32_t)((n + RT - 1) / RT);
    uint32_t probe_tiles = std::max<uint32_t>(total_tiles / probe_divisor(c, n, nq, k, !is_int), (uint32_t)(4 * k));
    probe_tiles = std::min<uint32_t>(std::min<uint32_t>(probe_tiles, total_tiles), (uint32_t)c->opt_probe_cap);
    // (the one-piece SQ8 bound is looser than the others: several times the candidates for the same probe)
    const size_t ccap = candidate_capacity(c, k, n, (size_t)probe_tiles * RT) * (is_sq8 ? 6 : 1);
    rc = ensure(c, c->cand, nqp * ccap * sizeof(uint2));
    if (rc) return rc;
    const bool have_tau = c->tau_override != nullptr;   // (retry pass: thresholds from the first pass's exact scores, no probe)
    if (have_tau)
        for (size_t q = 0; q < nq; q++) tau0[q] = c->tau_override[q];
    {
        // ONE upload per batch (vsgpu_mfma.hip): every per-query input of the filter is a region of ctx->qblock, staged in pinned memory
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t o_f2 = al(frag.size()), o_aux = o_f2 + al(frag32.size()), o_tau = o_aux + al(nqp * 4), o_cnt = o_tau + al(nqp * 4),
                     o_meta = o_cnt + al(nqp * 4), total = o_meta + al(qmeta.size() * 4);
        rc = ensure(c, c->qblock, total);
        if (rc) return rc;
        rc = ensure_pin_up(c, total);
        if (rc) return rc;
        char *hb = (char *)c->pin_up, *db = (char *)c->qblock.p;
        memcpy(hb, frag.data(), frag.size());
        if (!frag32.empty()) memcpy(hb + o_f2, frag32.data(), frag32.size());
        memcpy(hb + o_aux, qaux.data(), nqp * 4);
        memcpy(hb + o_tau, tau0.data(), nqp * 4);
        memset(hb + o_cnt, 0, nqp * 4);
        if (!qmeta.empty()) memcpy(hb + o_meta, qmeta.data(), qmeta.size() * 4);
        alias_into(c->qfrag, db, al(frag.size()));
        alias_into(c->qfrag2, db + o_f2, al(frag32.size()));
        alias_into(c->qn2, db + o_aux, al(nqp * 4));
        alias_into(c->tau, db + o_tau, al(nqp * 4));
        alias_into(c->counts, db + o_cnt, al(nqp * 4));
        alias_into(c->qmeta, db + o_meta, al(qmeta.size() * 4));
        rc = upload_block(c, db, hb, total);
        if (rc) return rc;
    }
    if (!is_int) {   // exact-order images for the re-rank, uploaded ahead of the scan (vsgpu_mfma.hip)
        rc = stage_queries(t, queries, nq, qstride);
        if (rc) return rc;
    }
    LowpParams P{};
    P.slabs = t->d_slabs;
    P.aux_slabs = (const uint32_t *const *)t->d_norm_slabs;
    P.slab_shift = t->slab_shift;
    P.slab_mask = (uint32_t)(((size_t)1 << t->slab_shift) - 1);
    P.row_stride = (uint32_t)t->row_bytes;
    P.n_rows = (uint32_t)n;
    P.qfrag = (const uint4 *)c->qfrag.p;
    P.qaux = (const uint32_t *)c->qn2.p;
    P.qmeta = (const float *)c->qmeta.p;
    if (is_sq8) {
        P.epi = t->metric == VSGPU_L2 ? LE_FP_L2 : LE_FP_IP;
        P.sq8_max = t->d_sq8_max;
        P.sq8_fmax = (float)(2.0 * 128.0 * 127.0 * (double)kdim * (1.0 + 1e-6));   // |D| <= 128 * 127 * width, |K| likewise
        P.sq8_ncmax = (float)(128.0 * std::sqrt((double)kdim) * 1.00001);
    } else if (is_int) {
        P.sq8_max = t->d_sq8_max;   // {min, max} of the rows' aux values (null for the 16-byte-record kinds)
        P.epi = t->epi == EPI_INT_L2 ? LE_I8_L2 : (t->epi == EPI_INT_IP ? (is_u8 ? LE_U8_IP : LE_I8_IP) : LE_I8_COS);
    } else {
        P.epi = t->metric == VSGPU_L2 ? LE_FP_L2 : LE_FP_IP;
        // bf16*bf16 / fp16*fp16 products are exact in fp32: only accumulation order/rounding differs
        const double u = std::ldexp(1.0, -24);
        const double cq = (double)kdim * std::ldexp(1.0, -22) * 1.01;
        const double gref = ((double)kdim / 16.0 + 12.0) * u;
        P.cE = (float)(((cq + 2.0 * gref + 4.0 * u) * 1.001) * (1.0 + 1e-6));
        P.absE = t->metric == VSGPU_L2 ? 1e-30f : 1e-6f;
    }
    P.tau = (const float *)c->tau.p;
    P.counts = (uint32_t *)c->counts.p;
    P.cand = (uint2 *)c->cand.p;
    P.cap = (uint32_t)ccap;

    const uint32_t tile_step = total_tiles / probe_tiles;
    // k_probe_threshold sorts M group minima per query in LDS; more probe tiles than that are grouped (the k-th
    // smallest group minimum still has k distinct rows at or below it, and with k << M grouping costs nothing)
    uint32_t M = 64;
    while (M < probe_tiles && M < 8192 && M < 64 * k) M <<= 1;   // (64 k groups: two of the k best rows rarely share one)
    rc = ensure(c, c->dense, nqp * (size_t)probe_tiles * 4);
    if (rc) return rc;
    // probe grid: as many workgroups as are resident at once (one per CU for these kernels) -- a second round of workgroups
    // loads its query fragments and fills its ring again while the CUs wait
    const uint32_t wgs = (uint32_t)c->n_cu * (uint32_t)c->opt_lowp_wg_per_cu;

    ScanChainGuard chain(t);   // behind the other reader lanes' probe + scan (see topk_mfma)
    if (c->opt_events & 2) HIPCHK(hipEventRecord(c->ev_c, c->stream));
    if (!have_tau) {
        LowpParams Q = P;
        Q.tile_first = 0;
        Q.tile_step = tile_step;
        Q.tile_run_shift = probe_run_shift(c, (size_t)RT * t->row_bytes, probe_tiles);
        Q.n_tiles = probe_tiles;
        Q.tilemin = (float *)c->dense.p;
        Q.tilemin_stride = probe_tiles;
        if (t->lp_wide) launch_wide_h16(t, MF_PROBE, wide_blocks, Q, dim3(std::min(probe_tiles, wgs * 2), (unsigned)q_tiles), c->stream);
        else if (qsplit) launch_lowp_i8_split(t, MF_PROBE, Q, dim3(std::min(probe_tiles, wgs), (unsigned)q_tiles), c->stream);
        else if (hsplit && t->lp_kind == LP_BF16) launch_lowp_h16_split<LP_BF16>(MF_PROBE, Q, dim3(std::min(probe_tiles, wgs), (unsigned)q_tiles), c->stream);
        else if (hsplit) launch_lowp_h16_split<LP_F16>(MF_PROBE, Q, dim3(std::min(probe_tiles, wgs), (unsigned)q_tiles), c->stream);
        else if (narrow) launch_lowp_narrow_any(t, MF_PROBE, Q, dim3(std::min(probe_tiles, wgs * 2), (unsigned)q_tiles), c->stream);
        else launch_lowp(t, MF_PROBE, Q, dim3(std::min(probe_tiles, wgs), (unsigned)q_tiles), c->stream);
        HIPCHK(hipGetLastError());
        rc = launch_probe_threshold(c, nq, probe_tiles, k, M);
        if (rc) return rc;
    }
    VSG_POLL_POINT(c);
    if (c->opt_events & 2) HIPCHK(hipEventRecord(c->ev_d, c->stream));
    chain.before_scan();   // behind the other lane's select kernel (ScanChain)
    if (c->opt_events & 1) HIPCHK(hipEventRecord(c->ev_a, c->stream));
    {
        LowpParams Q = P;
        Q.tile_first = 0;
        Q.tile_step = 1;
        Q.n_tiles = total_tiles;
        uint32_t fw = (uint32_t)c->n_cu * (uint32_t)c->opt_lowp_wg_per_cu;
        if (is_sq8 && !narrow && t->lp_ksteps < 16) fw = std::max(fw, (uint32_t)c->n_cu * 2u);   // (launch_lowp: the 128-register SQ8 filter)
        Q.dbg = (int)c->opt_lowp_dbg;
        uint32_t *d_ph = nullptr;
        const size_t ph_words = (size_t)fw * q_tiles * 16 * 8;
        if (Q.dbg & 8) {
            HIPCHK(hipMalloc(&d_ph, ph_words * 4));
            HIPCHK(hipMemsetAsync(d_ph, 0, ph_words * 4, c->stream));
            Q.tilemin = reinterpret_cast<float *>(d_ph);
        }
        if (x32) {
            Q.qfrag = (const uint4 *)c->qfrag2.p;
            int var = (int)c->opt_lowp_x32 - 1;
            if (c->opt_lowp_x32 == 1) {
                // Which 32 x 32 x 32 filter: the lean stream screens a unit of rows with ONE integer threshold per query, derived from
                // the table-wide extremes of the rows' aux values (norms / sums of squares) -- fine while those spread moderately (the
                // benchmark's uniform rows: min / max norm 0.857 over 50 M rows, the exact test then runs on a quarter of the
                // wave-units and the lean stream is 11 % ahead; it would lose once the test ran on ~90 % of them); a table whose rows
                // differ widely (min / max below 0.75), and a table nothing is known about yet, get the per-value screen.  The extremes
                // live on the device; a pinned host copy is refreshed BEHIND a batch whenever rows were added since, and read by the
                // next one -- stale values only understate the spread, and the choice affects speed alone, never a result.
                bool lean = Q.epi == LE_I8_IP;
                if (!lean && Q.epi != LE_U8_IP && t->h_i8_ext && t->i8_ext_n != (size_t)-1) {
                    const int lo = t->h_i8_ext[0], hi = t->h_i8_ext[1];
                    if (Q.epi == LE_I8_COS) {
                        float flo, fhi;
                        memcpy(&flo, &lo, 4);
                        memcpy(&fhi, &hi, 4);
                        lean = lo <= hi && flo > 0.0f && flo >= 0.75f * fhi;
                    } else lean = lo <= hi && lo > 0 && (double)lo >= 0.56 * (double)hi;
                }
                var = lean ? (262144 | 4 | 16384) : 32769;
                x32_lean = lean;
            } else x32_lean = (var & 262144) != 0;
            const uint32_t gx = std::min(total_tiles, (uint32_t)c->n_cu);
            uint64_t *d_clk = nullptr;
            const size_t clk_words = (size_t)gx * q_tiles * ((var & 512) ? X32_NW * 8 : 2);
            if (!(var & 262144) && (var & (256 | 512))) {   // (k_i8_filter_x32's diagnosis bits; the lean kernel's 256 is its per-unit threshold)   // diagnosis: shader clock = s_memtime ticks per 100 MHz s_memrealtime tick; phase sums
                HIPCHK(hipMalloc(&d_clk, clk_words * 8));
                HIPCHK(hipMemsetAsync(d_clk, 0, clk_words * 8, c->stream));
                Q.tilemin = reinterpret_cast<float *>(d_clk);
            }
            launch_i8_x32(t, var, Q, dim3(gx, (unsigned)q_tiles), c->stream);
            if (t->i8_ext_n != t->n && t->d_sq8_max) {   // (see above: the next batch's choice)
                if (!t->h_i8_ext) HIPCHK(hipHostMalloc((void **)&t->h_i8_ext, 16, hipHostMallocDefault));
                HIPCHK(hipMemcpyAsync(t->h_i8_ext, t->d_sq8_max, 8, hipMemcpyDeviceToHost, c->stream));
                t->i8_ext_n = t->n;
            }
            if (d_clk) {
                std::vector<uint64_t> h(clk_words);
                HIPCHK(hipMemcpyAsync(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(hipStreamSynchronize(c->stream));
                HIPCHK(hipFree(d_clk));
                if (var & 512) {
                    double sum[4] = {0, 0, 0, 0}, units = 0;
                    for (size_t w = 0; w < h.size() / 8; w++) {
                        for (int i = 0; i < 4; i++) sum[i] += (double)h[w * 8 + i];
                        units += (double)h[w * 8 + 4];
                    }
                    static int once = 0;
                    if (!once++) {
                        for (int wv = 0; wv < X32_NW; wv++) {
                            double ws[6] = {0, 0, 0, 0, 0, 0}, wu = 0;
                            for (size_t w = wv; w < h.size() / 8; w += X32_NW) {
                                for (int i = 0; i < 4; i++) ws[i] += (double)h[w * 8 + i];
                                ws[4] += (double)h[w * 8 + 5];
                                ws[5] += (double)h[w * 8 + 6];
                                wu += (double)h[w * 8 + 4];
                            }
                            if (wu > 0) fprintf(stderr, "  wave %d: top %.0f  requests %.0f  stream %.0f  tail %.0f  vmcnt %.0f  barrier %.0f\n", wv, ws[4] / wu, ws[0] / wu, ws[1] / wu, ws[5] / wu, ws[2] / wu, ws[3] / wu);
                        }
                    }
                    if (units > 0)
                        fprintf(stderr, "x32 phases, mean s_memtime ticks per wave and unit: requests %.0f  rest of the stream %.0f  vmcnt wait %.0f  barrier %.0f\n",
                                sum[0] / units, sum[1] / units, sum[2] / units, sum[3] / units);
                    h.resize(0);
                }
                double sc = 0, sr = 0;
                for (size_t i = 0; i < h.size(); i += 2) sc += (double)h[i], sr += (double)h[i + 1];
                if (!h.empty()) fprintf(stderr, "x32 clocks: %.0f shader ticks / %.0f ref ticks per workgroup = %.3f GHz (ref 100 MHz), %.3f ms\n",
                        sc / (h.size() / 2), sr / (h.size() / 2), sc / sr * 0.1, sr / (h.size() / 2) * 1e-5);
            }
        } else if (t->lp_wide) {
            // gridDim.x a multiple of 8: the query tiles of a row tile (blockIdx.y) land on one XCD and share its L2 (vsgpu_mfma.hip)
            const uint32_t per_cu = (wide_blocks >= 2 || KS > 192) ? 1u : 2u;   // (workgroups resident per CU)
            uint32_t gx = std::max<uint32_t>(8, std::min<uint32_t>(total_tiles, (uint32_t)c->n_cu * per_cu / (uint32_t)std::min<size_t>(q_tiles, 4)) / 8 * 8);
            if (c->opt_wide_gx > 0) gx = (uint32_t)c->opt_wide_gx;
            launch_wide_h16(t, MF_FILTER, wide_blocks, Q, dim3(std::min(total_tiles, gx), (unsigned)q_tiles), c->stream);
        } else if (qsplit) launch_lowp_i8_split(t, MF_FILTER, Q, dim3(std::min(total_tiles, fw), (unsigned)q_tiles), c->stream);
        else if (hsplit && t->lp_kind == LP_BF16) launch_lowp_h16_split<LP_BF16>(MF_FILTER, Q, dim3(std::min(total_tiles, fw), (unsigned)q_tiles), c->stream);
        else if (hsplit) launch_lowp_h16_split<LP_F16>(MF_FILTER, Q, dim3(std::min(total_tiles, fw), (unsigned)q_tiles), c->stream);
        else if (narrow) launch_lowp_narrow_any(t, MF_FILTER, Q, dim3(std::min(total_tiles, (uint32_t)c->n_cu * 2), (unsigned)q_tiles), c->stream);
        else if (!launch_lowp_variant(t, (int)c->opt_lowp_variant, Q, fw, (unsigned)q_tiles, c->stream))
            launch_lowp(t, MF_FILTER, Q, dim3(std::min(total_tiles, fw), (unsigned)q_tiles), c->stream);
        HIPCHK(hipGetLastError());
        if (d_ph) {  // phase sums of every wave: mean cycles per tile, printed once per launch
            std::vector<uint32_t> h(ph_words);
            HIPCHK(hipMemcpyAsync(h.data(), d_ph, ph_words * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipFree(d_ph));
            double sum[5] = {0, 0, 0, 0, 0}, tiles = 0;
            for (size_t w = 0; w < ph_words / 8; w++) {
                if (!h[w * 8 + 5]) continue;
                for (int i = 0; i < 5; i++) sum[i] += h[w * 8 + i];
                tiles += h[w * 8 + 5];
            }
            if (tiles > 0)
                fprintf(stderr, "lowp phases, mean s_memtime ticks per wave and tile: vmcnt-wait %.0f  barrier %.0f  refill-request %.0f  "
                                "reads+mfma-issue %.0f  epilogue %.0f\n",
                        sum[0] / tiles, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles, sum[4] / tiles);
        }
    }
    if (c->opt_events & 1) HIPCHK(hipEventRecord(c->ev_b, c->stream));
    // the scan is in the stream: the next reader lane's probe may follow it and run beside this lane's re-rank and select
    // kernels (small grids both); its SCAN waits for them (ScanChain)
    chain.scan_submitted_if_early();
    VSG_POLL_POINT(c);
    if (x32 && c->opt_lowp_x32 != 1 && (((int)c->opt_lowp_x32 - 1) & ((((int)c->opt_lowp_x32 - 1) & 262144) ? (64 | 128 | 512) : (32 | 64 | 128 | 512)))) {  // diagnosis variants of the 32x32x32 kernels: time only
        HIPCHK(hipStreamSynchronize(c->stream));
        account_scan(c, t, n, 1, "k_i8_filter_x32(dbg)");
        for (size_t q = 0; q < nq; q++) counts[q] = 0;
        return VSGPU_OK;
    }
    if (c->opt_lowp_dbg) {  // diagnosis run: the kernel's output is meaningless, report its time only
        HIPCHK(hipStreamSynchronize(c->stream));
        account_scan(c, t, n, 1, "k_mfma_filter_lowp(dbg)");
        for (size_t q = 0; q < nq; q++) counts[q] = 0;
        return VSGPU_OK;
    }
    if (!is_int) {
        rc = launch_exact_pairs(t, nq, ccap);
        if (rc) return rc;
    }
    return collect_candidates(t, queries, nq, qstride, k, cap, ccap, ids, scores, counts,
                              is_sq8 ? "k_mfma_filter_lowp(sq8)" : (t->lp_wide && is_int) ? "k_mfma_filter_wide(i8)" : t->lp_wide ? "k_mfma_filter_wide(h16)" : !is_int ? "k_mfma_filter_lowp(h16)"
                              : (tuning_ksplit_on(c) && KS == 16 && RT == 32 && !qsplit && !c->opt_lowp_variant) ? "k_i8_filter_ksplit"
                              : x32 ? (x32_lean ? "k_i8_filter_x32l" : "k_i8_filter_x32") : "k_mfma_filter_lowp(i8)", &chain);
}