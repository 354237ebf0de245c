// host_tier.h -- which reference ISA tier's summation order an index reproduces.
//
// The reference picks its distance kernel from the host CPU's features at run time (spaces.h:68-78
// getCpuOptimizationFeatures; choosers L2_space.cpp:185-516, IP_space.cpp:435-889), so the last bits of a
// reference reply depend on the machine it runs on.  A drop-in replacement follows the same rule: the tier comes from
// the host's CPUID --
//   avx512f                                  -> the AVX-512F / BW / VBMI2 / VNNI kernels' order     (VSGPU_TIER_AVX512)
//   ... && avx512_bf16 && avx512vl           -> vdpbf16ps first for bf16 IP / Cosine                (VSGPU_TIER_AVX512_BF16;
//                                               IP_space.cpp:585-590; every other type as AVX512)
//   no avx512f                               -> ALSO the AVX-512 kernels' order, with one line on stderr: a reference build
//                                               there runs its AVX2 / AVX / SSE kernels, whose orders are not restated
//                                               (DESIGN.md 3), so no tier this library has equals it; AVX512 keeps the MFMA
//                                               filters (the scalar order turns them off: 10-50x slower) and is the order
//                                               the same index gives on any AVX-512 host.  The scalar kernels' order
//                                               (VSGPU_TIER_SCALAR) is an explicit choice only.
//   ... && avx512_fp16 && avx512vl           -> STILL AVX512_BF16 (one line on stderr for fp16 indexes).  A reference built by gcc >= 12 /
//                                               clang >= 14 (OPT_AVX512_FP16_VL) accumulates fp16 rows of dim >= 32 in HALF precision
//                                               there (IP_space.cpp:649-658, L2_space.cpp:388-397); that order is restated
//                                               (VSGPU_TIER_AVX512_FP16) but OPT-IN: VECSIM_GPU_TIER=avx512_fp16 -- it is unpinned (no
//                                               host here executes it), a gcc-11 build has no such kernels, and it runs on the exact
//                                               kernels only.
// -- and VECSIM_GPU_TIER = avx512 | avx512_bf16 | avx512_fp16 | scalar overrides it.  The reference asks for more than avx512f per type (bf16: avx512bw && avx512vbmi2,
// L2_space.cpp:332-337; int8 / uint8: avx512bw && avx512vl && avx512vnni, L2_space.cpp:451-455; fp16: avx512bw && avx512vl):
// reference_order_missing() names what the host lacks for its own reference build to run the order this library restates for
// a type; index creation prints it once per type (VecSimGpu_HostTierNote returns the same text to callers and tests).
// VECSIM_GPU_HOST_FLAGS = comma-separated feature names replaces the CPUID probe (tests).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#if defined(__x86_64__)
#include <cpuid.h>
#endif

#include "vsgpu.h"

namespace vsa {

struct HostFeatures {
    bool avx512f = false, avx512bw = false, avx512vl = false, avx512vbmi2 = false, avx512vnni = false, avx512_bf16 = false;
    bool f16c = false, fma3 = false, avx = false;
    bool avx512_fp16 = false;
};

inline HostFeatures host_features() {
    HostFeatures f;
    if (const char *e = std::getenv("VECSIM_GPU_HOST_FLAGS")) {
        const std::string s = std::string(",") + e + ",";
        auto has = [&](const char *n) { return s.find(std::string(",") + n + ",") != std::string::npos; };
        f.avx512f = has("avx512f"), f.avx512bw = has("avx512bw"), f.avx512vl = has("avx512vl");
        f.avx512vbmi2 = has("avx512vbmi2"), f.avx512vnni = has("avx512vnni"), f.avx512_bf16 = has("avx512_bf16");
        f.f16c = has("f16c"), f.fma3 = has("fma3"), f.avx = has("avx");
        f.avx512_fp16 = has("avx512_fp16");
        return f;
    }
#if defined(__x86_64__)
    __builtin_cpu_init();
    f.avx512f = __builtin_cpu_supports("avx512f"), f.avx512bw = __builtin_cpu_supports("avx512bw");
    f.avx512vl = __builtin_cpu_supports("avx512vl"), f.avx512vbmi2 = __builtin_cpu_supports("avx512vbmi2");
    f.avx512vnni = __builtin_cpu_supports("avx512vnni"), f.avx512_bf16 = __builtin_cpu_supports("avx512bf16");
    f.f16c = __builtin_cpu_supports("f16c"), f.fma3 = __builtin_cpu_supports("fma"), f.avx = __builtin_cpu_supports("avx");
    {   // CPUID.(EAX=7,ECX=0):EDX[23] (gcc 11's __builtin_cpu_supports does not know the name); the OS state check is avx512f's
        unsigned a = 0, b = 0, c = 0, d = 0;
        if (f.avx512f && __get_cpuid_count(7, 0, &a, &b, &c, &d)) f.avx512_fp16 = (d >> 23) & 1u;
    }
#endif
    return f;
}

inline int tier_from_features(const HostFeatures &f) {
    // avx512_fp16 hosts are NOT promoted to VSGPU_TIER_AVX512_FP16 on their own (round 6, advisor): that order is the one tier whose
    // parity is unpinned (nothing here executes avx512_fp16), it only equals reference builds made by gcc >= 12 / clang >= 14, and it
    // turns the MFMA filters off for fp16 rows.  VECSIM_GPU_TIER=avx512_fp16 opts in; resolve_tier says so once on such a host.
    if (f.avx512f && f.avx512_bf16 && f.avx512vl) return VSGPU_TIER_AVX512_BF16;
    return VSGPU_TIER_AVX512;
}

// What a reference build on this host would need, beyond avx512f, to run the kernel order restated for `type` (VSGPU_F32 ...):
// the missing feature names, comma separated; empty when the host's own reference build runs the restated order (integers:
// every tier gives the same number).  "avx512f" alone when the host has no AVX-512 at all.
inline std::string reference_order_missing(const HostFeatures &f, int type) {
    if (!f.avx512f) return "avx512f";
    std::string m;
    auto need = [&](bool have, const char *n) {
        if (!have) m += (m.empty() ? "" : ",") + std::string(n);
    };
    if (type == VSGPU_BF16) need(f.avx512bw, "avx512bw"), need(f.avx512vbmi2, "avx512vbmi2");   // L2_space.cpp:332-337
    if (type == VSGPU_F16) need(f.avx512bw, "avx512bw"), need(f.avx512vl, "avx512vl");           // L2_space.cpp:391-409
    return m;
}

// `type` < 0: no per-type note (VecSimGpu_HostTier); otherwise index creation, one line on stderr per process and type when the
// host's own reference build would run other kernels than the ones whose order this library restates
inline int resolve_tier(int type = -1) {
    if (const char *e = std::getenv("VECSIM_GPU_TIER")) {
        if (!std::strcmp(e, "scalar")) return VSGPU_TIER_SCALAR;
        if (!std::strcmp(e, "avx512_bf16")) return VSGPU_TIER_AVX512_BF16;
        if (!std::strcmp(e, "avx512_fp16")) return VSGPU_TIER_AVX512_FP16;
        if (!std::strcmp(e, "avx512")) return VSGPU_TIER_AVX512;
    }
    const HostFeatures f = host_features();
    if (!f.avx512f) {
        static bool said = false;
        if (!said) {
            said = true;
            std::fprintf(stderr, "vecsim_amd: host CPU has no AVX-512: scores follow the reference's AVX-512 kernel order (its AVX2 / SSE "
                                 "orders are not restated); set VECSIM_GPU_TIER=scalar for the scalar kernels' order\n");
        }
    } else if (type >= 0 && type < 16) {
        static bool said_fp16 = false;
        if (type == VSGPU_F16 && f.avx512_fp16 && f.avx512vl && !said_fp16) {
            said_fp16 = true;
            std::fprintf(stderr, "vecsim_amd: host CPU has avx512_fp16: a reference built by gcc >= 12 / clang >= 14 accumulates fp16 rows of "
                                 "dim >= 32 in half precision here; this library keeps the AVX512F fp32-accumulate order (what a gcc-11 build "
                                 "runs) unless VECSIM_GPU_TIER=avx512_fp16 is set (that order is restated but unpinned, and runs without the "
                                 "MFMA filters)\n");
        }
        static bool said_for[16] = {};
        const std::string miss = reference_order_missing(f, type);
        if (!miss.empty() && !said_for[type]) {
            said_for[type] = true;
            std::fprintf(stderr, "vecsim_amd: host CPU lacks %s: a reference build here runs a lower tier for %s rows; scores follow the "
                                 "AVX-512 kernels' order (the lower tiers' orders are not restated)\n",
                         miss.c_str(), type == VSGPU_BF16 ? "bf16" : "fp16");
        }
    }
    return tier_from_features(f);
}

inline const char *tier_name(int tier) {
    return tier == VSGPU_TIER_SCALAR ? "SCALAR" : tier == VSGPU_TIER_AVX512_BF16 ? "AVX512_BF16" : tier == VSGPU_TIER_AVX512_FP16 ? "AVX512_FP16" : "AVX512";
}

}  // namespace vsa
