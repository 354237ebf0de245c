/*
 * vso.h -- CPU ORACLE for the VecSim distance-kernel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and only as
 * the checker.  The product (vectorsimilarity_amd/) never links or imports it.
 *
 * What it restates (reference = RedisAI/VectorSimilarity, paths relative to src/VecSim/):
 *   - scalar kernels              spaces/L2/L2.cpp:76-174, spaces/IP/IP.cpp:185-286
 *   - AVX-512 tier kernels        spaces/L2/L2_AVX512F_FP32.h:21-59, IP/IP_AVX512F_FP32.h:19-56,
 *                                 L2/L2_AVX512F_FP64.h:21-59 (+IP twin), L2|IP/..._AVX512F_FP16.h,
 *                                 L2/L2_AVX512BW_VBMI2_BF16.h:40-78, IP/IP_AVX512BW_VBMI2_BF16.h,
 *                                 IP/IP_AVX512_BF16_VL_BF16.h:14-47, ..._VNNI_INT8.h / _UINT8.h
 *   - tier choice                 spaces/L2_space.cpp:185-516, spaces/IP_space.cpp:435-889
 *   - normalisation               spaces/normalize/normalize_naive.h:24-88, compute_norm.h:18-31
 *   - bf16 / fp16 conversions     types/bfloat16.h:23-39, types/float16.h:33-117
 *   - Flat top-K / range scans    algorithms/brute_force/brute_force.h:242-326
 *
 * Parity pin: checked against the reference's own known-answer tests (tests/unit/test_spaces.cpp,
 * test_bruteforce.cpp, test_int8.cpp ... restated as data in tests/golden/) by tests/test_oracle*.py, and -- the scalar
 * kernels, the conversions, normalisation, scalar SQ8, the top-k containers and heaps -- against bits the reference ITSELF
 * produced: oracle/_ref is the reference's own scalar translation units compiled where they lie (oracle/build_ref.sh,
 * tests/golden/ref_scalar_random.json).  The SIMD kernel headers cannot be compiled in the build image without writing
 * stand-in headers for the un-vendored cpu_features dependency (every one includes it through
 * spaces/space_includes.h:13-16): the "AVX-512 order" variants are restated from the source lines above and cross-checked
 * against an independent AVX-512 intrinsics implementation of the same published algorithm (vso_fast.c) on the host CPU.
 * PARITY UNPINNED for one tier: VSO_TIER_AVX512_FP16 (half-precision accumulators, gcc >= 12 builds on avx512_fp16 hosts) --
 * no toolchain or CPU within reach emits or executes it; see vso.c.
 */
#ifndef VSO_H
#define VSO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* numeric values match VecSimType / VecSimMetric (vec_sim_common.h:60-69,87) */
enum { VSO_F32 = 0, VSO_F64 = 1, VSO_BF16 = 2, VSO_F16 = 3, VSO_I8 = 4, VSO_U8 = 5 };
enum { VSO_L2 = 0, VSO_IP = 1, VSO_COSINE = 2 };
/* arithmetic profile = which ISA tier of the reference is being restated */
enum {
    VSO_TIER_AVX512 = 0, /* gcc-11 build on an AVX-512F/BW/VL/VNNI/VBMI2 host, avx512_bf16 masked off */
    VSO_TIER_SCALAR = 1, /* "no optimisation" kernels */
    VSO_TIER_AVX512_BF16 = 2, /* as AVX512 plus the vdpbf16ps tier for bf16 IP */
    VSO_TIER_AVX512_FP16 = 3 /* as AVX512_BF16 plus the half-accumulating kernels for fp16 rows of dim >= 32 (gcc >= 12 builds on
                                avx512_fp16 hosts); parity UNPINNED: restated from the source alone, see vso.c */
};
/* IEEE half arithmetic, one rounding to nearest-even each (what vfmadd...ph / vmulph / vaddph / vsubph compute) */
uint16_t vso_h_fma(uint16_t a, uint16_t b, uint16_t c);
uint16_t vso_h_mul(uint16_t a, uint16_t b);
uint16_t vso_h_add(uint16_t a, uint16_t b);
uint16_t vso_h_sub(uint16_t a, uint16_t b);

size_t vso_elem_size(int type);
/* bytes of one stored row / query blob: dim*elem (+4 for int8/uint8 Cosine)   vec_utils.cpp:296-302 */
size_t vso_blob_size(int type, int metric, size_t dim);

/* one distance, result widened to double (float results widen exactly) */
double vso_distance(int type, int metric, int tier, size_t dim, const void *a, const void *b);
/* 1 when the (type,metric,dim,tier) combination resolves to the scalar kernel */
int vso_uses_scalar(int type, int metric, int tier, size_t dim);

/* n distances query-vs-rows; rows are `stride` bytes apart */
void vso_scan(int type, int metric, int tier, size_t dim, const void *rows, size_t n, size_t stride,
              const void *query, double *out);

/* in-place normalisation exactly as VecSim_Normalize (vec_sim.cpp:238-254) */
void vso_normalize(void *blob, size_t dim, int type);

/* conversions (bit-exact restatements) */
uint16_t vso_f32_to_bf16(float f);
float vso_bf16_to_f32(uint16_t h);
uint16_t vso_f32_to_f16(float f);
float vso_f16_to_f32(uint16_t h);
void vso_f32_to_bf16_n(const float *in, size_t n, uint16_t *out);
void vso_f32_to_f16_n(const float *in, size_t n, uint16_t *out);
void vso_bf16_to_f32_n(const uint16_t *in, size_t n, float *out);
void vso_f16_to_f32_n(const uint16_t *in, size_t n, float *out);

/* Flat top-K with the reference's sequential heap semantics (brute_force.h:257-288).
 * scores[i] belongs to internal id i, labels[i] is its label (NULL => label == id).
 * Writes min(k,n) results ascending by (score,label); returns the count. */
size_t vso_topk_replay(const double *scores, const size_t *labels, size_t n, size_t k,
                       size_t *out_labels, double *out_scores);
/* multi-value variant (labels may repeat; brute_force_multi.h + utils/updatable_heap.h) */
size_t vso_topk_replay_multi(const double *scores, const size_t *labels, size_t n, size_t k,
                             size_t *out_labels, double *out_scores);
/* range scan: score <= radius, in id order (brute_force.h:305-318); returns count */
size_t vso_range_replay(const double *scores, const size_t *labels, size_t n, double radius,
                        size_t *out_labels, double *out_scores);

/* ---- SQ8: uint8 codes + FP32 metadata (vso_sq8.c; types/sq8.h:19-62, preprocessors.h:259-649, IP.cpp:34-183,
 * L2.cpp:30-45,185-201 and their AVX-512 twins).  `metric` is the index metric; Cosine blobs are normalised before. ---- */
size_t vso_sq8_storage_size(int metric, size_t dim); /* dim + 12 (IP/Cosine) or 16 (L2) bytes */
size_t vso_sq8_query_size(int metric, size_t dim);   /* (dim + 1 or 2) floats */
void vso_sq8_quantize(const float *x, size_t dim, int metric, uint8_t *out);
void vso_sq8_query_blob(const float *y, size_t dim, int metric, float *out);
int vso_sq8_fp32_uses_scalar(int tier, size_t dim);
int vso_sq8_sq8_uses_scalar(int tier, size_t dim);
double vso_sq8_fp32_distance(int metric, int tier, size_t dim, const void *storage, const void *query);
double vso_sq8_sq8_distance(int metric, int tier, size_t dim, const void *a, const void *b);
void vso_sq8_fp32_scan(int metric, int tier, size_t dim, const void *rows, size_t n, size_t stride, const void *query,
                       double *out);
/* FP16 inputs / queries (QuantPreprocessor<float16>, SQ8_FP16_*): query blob = dim fp16 values + FP32 metadata */
/* mean-centred blobs (QuantPreprocessor<..., WithNorm = true> + DistanceCalculatorWithNorm; metric L2 or IP) */
size_t vso_sq8_storage_size_norm(int metric, size_t dim);
size_t vso_sq8_query_size_norm(int metric, size_t dim);
size_t vso_sq8_query_size_norm_f16(int metric, size_t dim);
void vso_sq8_quantize_norm(const float *x, const float *mean, size_t dim, int metric, uint8_t *out);
void vso_sq8_quantize_norm_f16(const uint16_t *x, const float *mean, size_t dim, int metric, uint8_t *out);
void vso_sq8_query_blob_norm(const float *y, const float *mean, size_t dim, int metric, float *out);
void vso_sq8_query_blob_norm_f16(const uint16_t *y, const float *mean, size_t dim, int metric, void *out);
double vso_sq8_fp32_distance_norm(int metric, int tier, size_t dim, const void *storage, const void *query);
double vso_sq8_fp16_distance_norm(int metric, int tier, size_t dim, const void *storage, const void *query);
double vso_sq8_sq8_distance_norm(int metric, int tier, size_t dim, const void *a, const void *b, float mean_sum_squares);
void vso_sq8_fp32_scan_norm(int metric, int tier, size_t dim, const void *rows, size_t n, size_t stride, const void *query,
                            double *out);
void vso_sq8_fp16_scan_norm(int metric, int tier, size_t dim, const void *rows, size_t n, size_t stride, const void *query,
                            double *out);
size_t vso_sq8_query_size_f16(int metric, size_t dim);
void vso_sq8_quantize_f16(const uint16_t *x, size_t dim, int metric, uint8_t *out);
void vso_sq8_query_blob_f16(const uint16_t *y, size_t dim, int metric, void *out);
int vso_sq8_fp16_uses_scalar(int tier, size_t dim);
double vso_sq8_fp16_distance(int metric, int tier, size_t dim, const void *storage, const void *query);
void vso_sq8_fp16_scan(int metric, int tier, size_t dim, const void *rows, size_t n, size_t stride, const void *query,
                       double *out);

/* whole Flat query on host rows (scan + replay) */
size_t vso_flat_topk(int type, int metric, int tier, size_t dim, const void *rows, size_t n,
                     size_t stride, const size_t *labels, const void *query, size_t k,
                     size_t *out_labels, double *out_scores);

/* ---- HNSW query loops over a given graph (vso_hnsw.c; hnsw.h:530-613, 1210-1258, 1967-2084) ----
 * graph layout = what VecSimGpu_HnswGraphCopy exports.  Returns the number of results (<= k),
 * ascending (score, label). */
size_t vso_hnsw_search(int type, int metric, int tier, size_t dim, const void *rows, size_t stride, uint32_t n,
                       const uint32_t *links0, const uint16_t *cnt0, uint32_t M0, const uint32_t *upper_off,
                       const uint32_t *upper, uint32_t M, const uint8_t *deleted, const uint64_t *labels,
                       uint32_t entry, int max_level, const void *query, size_t k, size_t ef, uint64_t *out_labels,
                       double *out_scores, uint64_t *dist_evals);
/* the same over a multi-value index (labels repeat): top_candidates is the label-keyed heap of hnsw_multi.h:108-112 */
size_t vso_hnsw_search_multi(int type, int metric, int tier, size_t dim, const void *rows, size_t stride, uint32_t n,
                       const uint32_t *links0, const uint16_t *cnt0, uint32_t M0, const uint32_t *upper_off,
                       const uint32_t *upper, uint32_t M, const uint8_t *deleted, const uint64_t *labels,
                       uint32_t entry, int max_level, const void *query, size_t k, size_t ef, uint64_t *out_labels,
                       double *out_scores, uint64_t *dist_evals);
/* range search (hnsw.h:616-680, 2087-2187): results in discovery order, returns their number (may exceed
 * out_cap, only out_cap are stored) */
size_t vso_hnsw_range(int type, int metric, int tier, size_t dim, const void *rows, size_t stride, uint32_t n,
                      const uint32_t *links0, const uint16_t *cnt0, uint32_t M0, const uint32_t *upper_off,
                      const uint32_t *upper, uint32_t M, const uint8_t *deleted, const uint64_t *labels,
                      uint32_t entry, int max_level, const void *query, double radius, double epsilon,
                      uint64_t *out_labels, double *out_scores, size_t out_cap, uint64_t *dist_evals);
/* batch iterator (hnsw_batch_iterator.h:96-230 + the single / multi heaps): a whole iteration in one call, batch b asking for
 * sizes[b] results; returns the batches taken, their results one after the other (out_counts[b] each) */
size_t vso_hnsw_iterate(int type, int metric, int tier, size_t dim, const void *rows, size_t stride, uint32_t n,
                        const uint32_t *links0, const uint16_t *cnt0, uint32_t M0, const uint32_t *upper_off,
                        const uint32_t *upper, uint32_t M, const uint8_t *deleted, const uint64_t *labels,
                        uint32_t entry, int max_level, const void *query, size_t ef, int multi, size_t n_labels,
                        const size_t *sizes, size_t max_batches, uint64_t *out_labels, double *out_scores, size_t *out_counts,
                        int *depleted_out);


/* ---- HNSW insert path (vso_hnsw.c; hnsw.h:418-422, 743-963, 1567-1610, 1857-1946): serial inserts in id order.  Pinned by the
 * reference-built graphs of tests/golden/ref_hnsw_graphs.npz.  Layout of the outputs: the index's graph export. */
void vso_hnsw_levels(uint32_t n, uint32_t M, uint32_t seed, uint8_t *levels);
uint64_t vso_hnsw_build(int type, int metric, int tier, size_t dim, const void *rows, size_t stride, uint32_t n, uint32_t M, uint32_t efc,
                        uint32_t seed, int fast, uint32_t *links0, uint16_t *cnt0, uint8_t *levels, uint32_t *upper_off, uint32_t *upper,
                        uint32_t *entry_out, int *max_level_out);

/* ---- timing leg (bench.py cpu_baseline, kind "port") ----
 * Same arithmetic as VSO_TIER_AVX512, written with AVX-512 intrinsics when the host has them
 * (falls back to the portable lanes code otherwise).  nq queries, `threads` OpenMP threads, one
 * query per thread at a time (mirrors bindings.cpp:250-283).  Returns 1 if the intrinsics path ran. */
int vso_flat_topk_batch_fast(int type, int metric, size_t dim, const void *rows, size_t n,
                             size_t stride, const void *queries, size_t nq, size_t qstride,
                             size_t k, int threads, size_t *out_labels, double *out_scores);
/* the same with the tier spelled out: every (type, metric, tier) of vso_distance has an intrinsics twin in vso_fast.c
 * (fp64, fp16, bf16 VBMI2, bf16 vdpbf16ps, int8 / uint8 VNNI); vso_fast_available says whether it can run on this host at this dim */
int vso_flat_topk_batch_fast_tier(int type, int metric, int tier, size_t dim, const void *rows, size_t n,
                                  size_t stride, const void *queries, size_t nq, size_t qstride,
                                  size_t k, int threads, size_t *out_labels, double *out_scores);
/* scores of nq queries x n rows (out[q * n + i]), row blocks dealt over `threads` threads: the full-size checker's leg */
int vso_scan_batch_fast_tier(int type, int metric, int tier, size_t dim, const void *rows, size_t n, size_t stride,
                             const void *queries, size_t nq, size_t qstride, int threads, double *out);
int vso_fast_available(int type, int metric, int tier, size_t dim);
double vso_distance_fast(int type, int metric, size_t dim, const void *a, const void *b);   /* tier AVX512 */
double vso_distance_fast_tier(int type, int metric, int tier, size_t dim, const void *a, const void *b);
int vso_has_avx512(void);
/* fp16 F16C tier (L2_F16C_FP16.h / IP_F16C_FP16.h): the portable restatement at any dim >= 8, and the same
 * algorithm on the host's F16C/FMA units (NaN when absent) -- the tests compare them bit for bit */
double vso_f16c_distance(int metric, size_t dim, const void *a, const void *b);
double vso_f16c_distance_hw(int metric, size_t dim, const void *a, const void *b);
int vso_has_f16c(void);

/* deterministic synthetic data shared with the device generator (vsgpu_fill_uniform):
 * value(seed, idx) is a pure function, so any row can be re-created on the host. */
uint32_t vso_hash32(uint64_t seed, uint64_t idx);
float vso_synth_f32(uint64_t seed, uint64_t idx);           /* U[-1,1) */
void vso_synth_rows_f32(uint64_t seed, uint64_t first_elem, size_t count, float *out);
int8_t vso_synth_i8(uint64_t seed, uint64_t idx);           /* uniform [-128,127] */

#ifdef __cplusplus
}
#endif
#endif
