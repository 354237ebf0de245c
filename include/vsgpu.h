/*
 * vsgpu.h -- the thin C-ABI shim between the C++ host index code and the HIP (gfx950) kernels.
 *
 * Everything that touches the GPU goes through these entry points: plain pointers, sizes and POD
 * structs only -- no C++ types, no torch types.  The host library (libvecsim_amd.so, C API in
 * VecSim/vec_sim.h) is the only in-tree caller; a reference maintainer would bind the same entry
 * points from DistanceCalculatorCommon / BruteForceIndex.
 *
 * What each group replaces in the reference (paths relative to src/VecSim/):
 *   table_*      the device mirror of DataBlocksContainer          containers/data_blocks_container.h:18-47,
 *                (rows packed `row_bytes` apart, ids dense [0,n))   containers/data_block.cpp:13-36
 *   topk         BruteForceIndex::topKQuery's scan + score filter   algorithms/brute_force/brute_force.h:242-291
 *   range        BruteForceIndex::rangeQuery's scan                 brute_force.h:293-326
 *   scores       BFS_BatchIterator::calculateScores / calcDistance  brute_force/bfs_batch_iterator.h:24-41,
 *                                                                   vec_sim_index.h:175-190
 * The arithmetic of every distance is the reference's AVX-512 tier (or its scalar tier below the
 * choosers' minimum dims): spaces/L2_space.cpp:185-516, spaces/IP_space.cpp:435-889.
 */
#ifndef VSGPU_H
#define VSGPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* element types / metrics: numeric values equal VecSimType / VecSimMetric (vec_sim_common.h:60-69,87) */
enum { VSGPU_F32 = 0, VSGPU_F64 = 1, VSGPU_BF16 = 2, VSGPU_F16 = 3, VSGPU_I8 = 4, VSGPU_U8 = 5,
       /* SQ8 storage rows (types/sq8.h:19-62): dim uint8 codes + FP32 {min, delta, sum[, sum_squares for L2]}, scored
        * against FP32 query blobs {y[dim], y_sum[, y_sum_squares]} by the asymmetric kernels of IP.cpp:34-80,
        * L2.cpp:30-45 and their AVX-512 twins.  Not a VecSimType: reached through VecSimGpu_NewFlatSQ8. */
       VSGPU_SQ8 = 6,
       /* the same SQ8 rows scored against FP16 query blobs {y[dim] fp16, y_sum[, y_sum_squares] FP32}: SQ8_FP16_*
        * (IP.cpp:82-144, L2.cpp:47-74, IP_AVX512F_SQ8_FP16.h: four 16-lane accumulators, scalar below dim 16) */
       VSGPU_SQ8H = 7 };
enum { VSGPU_L2 = 0, VSGPU_IP = 1, VSGPU_COSINE = 2 };
/* which reference ISA tier's summation order the kernels reproduce */
/* AVX512_FP16: as AVX512_BF16, and fp16 rows of dim >= 32 in the order of the half-ACCUMULATING kernels a gcc >= 12 build runs on an
 * avx512_fp16 host (IP_AVX512FP16_VL_FP16.h:16-51, L2_AVX512FP16_VL_FP16.h:16-58): exact kernels only, no MFMA filter */
enum { VSGPU_TIER_AVX512 = 0, VSGPU_TIER_SCALAR = 1, VSGPU_TIER_AVX512_BF16 = 2, VSGPU_TIER_AVX512_FP16 = 3 };

enum {
    VSGPU_OK = 0,
    VSGPU_ERR_NO_DEVICE = 1,   /* no HIP device / runtime failure: the product never falls back to CPU */
    VSGPU_ERR_HIP = 2,
    VSGPU_ERR_ARG = 3,
    VSGPU_ERR_UNSUPPORTED = 4,
    VSGPU_ERR_OOM = 5,
    VSGPU_ERR_TIMEOUT = 6      /* the caller's poll function asked to stop (vsgpu_set_poll) */
};

typedef struct vsgpu_ctx vsgpu_ctx;     /* one per (process, device): stream, scratch, timers */
typedef struct vsgpu_table vsgpu_table; /* device-resident rows of one Flat index */

/* ---- runtime ---- */
int vsgpu_device_count(void);                 /* 0 when no GPU is visible */
int vsgpu_device_synchronize(int device);     /* hipDeviceSynchronize on `device`: every stream of every context (bench brackets) */
const char *vsgpu_last_error(void);           /* thread-local message of the last failure */
vsgpu_ctx *vsgpu_ctx_create(int device);      /* NULL on failure (see vsgpu_last_error) */
void vsgpu_ctx_destroy(vsgpu_ctx *ctx);
int vsgpu_ctx_device(const vsgpu_ctx *ctx);
int vsgpu_ctx_sync(vsgpu_ctx *ctx);

/* ---- device table (mirror of DataBlocksContainer) ---- */
/* row_bytes = storedDataSize (dim*sizeof(T), +4 for int8/uint8 Cosine: utils/vec_utils.cpp:296-302).
 * `metric` here is the *kernel* metric: fp Cosine indexes pass VSGPU_IP (spaces.cpp:27-30,53-56). */
vsgpu_table *vsgpu_table_create(vsgpu_ctx *ctx, int type, int metric, int tier, size_t dim,
                                size_t row_bytes);
void vsgpu_table_destroy(vsgpu_table *t);
size_t vsgpu_table_size(const vsgpu_table *t);
size_t vsgpu_table_bytes(const vsgpu_table *t);          /* device bytes held */
int vsgpu_table_append(vsgpu_table *t, const void *host_rows, size_t n);  /* ids n_old .. n_old+n-1 */
int vsgpu_table_write(vsgpu_table *t, size_t id, const void *host_row);   /* updateElement */
int vsgpu_table_move(vsgpu_table *t, size_t dst_id, size_t src_id);       /* swap-delete copy */
int vsgpu_table_truncate(vsgpu_table *t, size_t new_size);
int vsgpu_table_read(vsgpu_table *t, size_t id, void *host_row);          /* getElement */
int vsgpu_table_read_range(vsgpu_table *t, size_t first, size_t n, void *host_rows);   /* rows [first, first + n) as stored */
/* Reader lanes: a view shares the parent's rows but runs its queries on another context (own stream and scratch), so
 * concurrent readers -- which the reference allows on one index (vec_sim.h, bindings.cpp:250-283 knn_parallel) -- overlap
 * one reader's small kernels, copies and host work with another reader's scan kernel.  The table-wide scan kernels of all
 * lanes are chained on the GPU in submission order.  The caller keeps writers out while views are queried and calls
 * _view_sync after the parent's rows changed; vsgpu_table_destroy frees a view (before its parent). */
vsgpu_table *vsgpu_table_view_create(vsgpu_table *parent, vsgpu_ctx *ctx);
int vsgpu_table_view_sync(vsgpu_table *view);
/* append n synthetic rows generated on the device: element j of row i is
 * synth(seed, (first_id+i)*dim + j), bit-identical to oracle/vso.c:vso_synth_f32 (fp32 only) */
int vsgpu_table_append_synthetic(vsgpu_table *t, size_t n, uint64_t seed);

/* ---- queries ----
 * `queries`: nq preprocessed query blobs, `qstride` bytes apart, on the HOST (borrowed for the call).
 *
 * vsgpu_topk: for every query returns every row whose score <= T_q, T_q = the k-th smallest exact
 * score over the table (all rows when size < k), ascending by internal id, with the exact
 * (reference-order) score widened to double.  That set is what the sequential heap of
 * brute_force.h:264-281 needs to reproduce the reference result.
 *   ids/scores: [nq][cap]; counts[q] = number written, or VSGPU_COUNT_OVERFLOW when more than cap
 *   rows tie at or below T_q (caller then uses vsgpu_scores for that query). */
#define VSGPU_COUNT_OVERFLOW 0xFFFFFFFFu
int vsgpu_topk(vsgpu_table *t, const void *queries, size_t nq, size_t qstride, size_t k,
               size_t cap, uint32_t *ids, double *scores, uint32_t *counts);
/* rows with score <= radius, ascending id (brute_force.h:305-318). Same overflow convention. */
int vsgpu_range(vsgpu_table *t, const void *query, double radius, size_t cap, uint32_t *ids,
                double *scores, uint32_t *count);
/* ---- device-resident score vector: the Flat batch iterator's state (bfs_batch_iterator.h:24-41 materialises
 * all n scores on the host; here they stay in HBM).  _next returns every not-yet-retired row whose score is at or
 * below the k-th smallest live score (count = VSGPU_COUNT_OVERFLOW when more than cap rows qualify), _retire
 * removes rows from later calls, _read copies all n scores (NaN for retired rows) to the host. */
typedef struct vsgpu_scorebuf vsgpu_scorebuf;
vsgpu_scorebuf *vsgpu_scorebuf_create(vsgpu_table *t, const void *query); /* NULL: fp64 table, empty table, no memory */
void vsgpu_scorebuf_destroy(vsgpu_scorebuf *b);
size_t vsgpu_scorebuf_rows(const vsgpu_scorebuf *b);
int vsgpu_scorebuf_next(vsgpu_scorebuf *b, size_t k, size_t cap, uint32_t *ids, double *scores, uint32_t *count);
int vsgpu_scorebuf_retire(vsgpu_scorebuf *b, const uint32_t *rows, size_t m);
int vsgpu_scorebuf_read(vsgpu_scorebuf *b, double *all);
/* dense exact scores of rows [first, first+n) against one query */
int vsgpu_scores(vsgpu_table *t, const void *query, size_t first, size_t n, double *scores);
/* exact scores of an explicit list of rows against one query (getDistanceFrom / ad-hoc BF) */
int vsgpu_scores_of(vsgpu_table *t, const void *query, const uint32_t *ids, size_t n,
                    double *scores);

/* SQ8 tables only: symmetric distances (SQ8_SQ8_InnerProduct / _Cosine / _L2Sqr, IP.cpp:146-183, L2.cpp:185-201 and the
 * AVX-512 VNNI twins) between stored rows ids_a[i] and ids_b[i]. */
int vsgpu_sq8_pair_scores(vsgpu_table *t, const uint32_t *ids_a, const uint32_t *ids_b, size_t n, double *scores);
/* Mean-centred SQ8 (QuantPreprocessor<..., WithNorm = true> + DistanceCalculatorWithNorm, calculator.h:126-232).  L2 tables
 * need nothing: blobs keep their layout and the base kernels give the distance.  An IP table created with row_bytes = dim + 16
 * holds rows {codes, min, delta, sum, x_mean_ip}; its query blobs are {y[dim], y_sum, y_mean_ip} and every score is
 * base - y_mean_ip; pair scores are base - x_mean_ip - y_mean_ip + mean_sum_squares with the constant set here. */
int vsgpu_table_set_sq8_mean_sum_squares(vsgpu_table *t, float mean_sum_squares);
/* SQ8 tables: extremes of the stored rows' metadata {max delta, min delta, max min, min min, max |codes - 128|_2, min sum_squares,
 * max sum_squares, 0}.  Rounds 1-2 used them for a block pre-screen in the MFMA filter; since round 3 the filter's per-value
 * screen (mfma_lowp_kernels.hpp epilogue_sq8) takes the table-wide maxima it needs from the device-side aux pass
 * (k_row_aux_sq8), so the call is accepted, kept with the table and has no effect on any result or kernel. */
int vsgpu_table_set_sq8_block_bounds(vsgpu_table *t, const float bounds[8]);

/* ---- HNSW query loops (algorithms/hnsw/hnsw.h:530-613, 1210-1258, 1967-2084) ----
 * A device snapshot of the graph the host index built (vectors stay in the vsgpu_table): level-0
 * adjacency [n][2M] + counts, upper-level blocks, deletion flags, labels, entry point.  The search
 * kernel replays the reference's greedy descent + ef-bounded best-first search, one wavefront per
 * query, scoring neighbours with the same exact kernels as the Flat path. */
typedef struct vsgpu_graph vsgpu_graph;
vsgpu_graph *vsgpu_graph_create(vsgpu_table *t, size_t M);
void vsgpu_graph_destroy(vsgpu_graph *g);
/* a second reader of the same snapshot: `view_table` is a view (vsgpu_table_view_create) of the graph's table on another context;
 * searches through the view use that context's stream and scratch and their own visited tags (no upload through a view) */
vsgpu_graph *vsgpu_graph_view_create(vsgpu_graph *parent, vsgpu_table *view_table);
/* multi-value index (hnsw_multi.h): labels repeat across nodes, top_candidates keeps one entry per label (its lowest distance) */
void vsgpu_graph_set_multi(vsgpu_graph *g, int multi);
/* upper_off[i]: index (in blocks of 1+M words) of node i's level-1 block inside `upper`, 0xFFFFFFFF
 * for level-0-only nodes; block l-1 of a node holds {count, links[M]} of level l. */
int vsgpu_graph_upload(vsgpu_graph *g, size_t n, const uint32_t *links0, const uint16_t *cnt0,
                       const uint32_t *upper_off, const uint32_t *upper, size_t upper_words,
                       const uint8_t *deleted, const uint64_t *labels, uint32_t entry, int max_level);
/* labels/scores are [nq][k] (ascending score, then label); counts[q] <= k results written.
 * `dist_evals` (may be NULL) receives the number of distance evaluations of the call. */
int vsgpu_graph_search(vsgpu_graph *g, const void *queries, size_t nq, size_t qstride, size_t k, size_t ef,
                       uint64_t *labels, double *scores, uint32_t *counts, uint64_t *dist_evals);
/* Range search (hnsw.h:2087-2187, processCandidate_RangeSearch hnsw.h:616-680): labels/scores are [nq][cap] in
 * discovery order.  counts[q] bits 0..30 = results found (only min(found, cap) are stored: call again with a
 * larger cap when it is exceeded); bit 31 = the candidate window (LDS) overflowed, the list may be incomplete. */
int vsgpu_graph_range(vsgpu_graph *g, const void *queries, size_t nq, size_t qstride, double radius, double epsilon,
                      size_t cap, uint64_t *labels, double *scores, uint32_t *counts, uint64_t *dist_evals);

/* ---- shard exchange of a multi-GPU Flat index: RCCL over xGMI, one process per GPU ----
 * The reference has no counterpart (it is single-process, CPU only); the semantics the exchange must
 * preserve are those of BruteForceIndex::topKQuery over the union of the shards (brute_force.h:242-291).
 * Rank 0 draws a 128-byte id (ncclGetUniqueId) and hands it to the other ranks out of band; every rank then
 * creates its communicator.  allgather: every rank contributes `bytes` HOST bytes and receives world*bytes
 * in rank order; broadcast: `buf` of the root reaches every rank.  Both return when the data is in place. */
#define VSGPU_COMM_ID_BYTES 128
typedef struct vsgpu_comm vsgpu_comm;
int vsgpu_comm_unique_id(void *id128);
vsgpu_comm *vsgpu_comm_create(vsgpu_ctx *ctx, int rank, int world, const void *id128);
void vsgpu_comm_destroy(vsgpu_comm *c);
int vsgpu_comm_rank(const vsgpu_comm *c);
int vsgpu_comm_world(const vsgpu_comm *c);
int vsgpu_comm_allgather(vsgpu_comm *c, const void *send, size_t bytes, void *recv);
int vsgpu_comm_broadcast(vsgpu_comm *c, void *buf, size_t bytes, int root);
/* A collective that fails on a rank (RCCL / HIP error, or the peers not arriving within $VECSIM_GPU_EXCHANGE_TIMEOUT_MS, default
 * 120 000) aborts that rank's communicator (ncclCommAbort) before it returns the error, so that its peers' collectives end -- by
 * the asynchronous error or by their own time limit -- instead of waiting for ever; later calls on it are refused.
 * vsgpu_comm_abort does the same on the caller's decision.  vsgpu_comm_staged: 1 = device send / receive buffers with copies to
 * and from the mapped host blocks (default for world > 1), 0 = the mapped host blocks are the collective's buffers
 * ($VECSIM_GPU_EXCHANGE = staged | mapped overrides). */
int vsgpu_comm_abort(vsgpu_comm *c);
int vsgpu_comm_staged(const vsgpu_comm *c);

/* ---- measurement hooks (bench.py roofline leg) ----
 * HIP-event time of the dominant scan kernel, accumulated per ctx on the stream it runs on. */
typedef struct {
    double scan_ms;        /* sum of scan-kernel durations since the last reset */
    uint64_t scan_launches;
    uint64_t scan_rows;    /* rows streamed by those launches */
    uint64_t scan_bytes;   /* rows * row_bytes (the algorithmic bytes of SURVEY.md §8d) */
    double other_ms;       /* probe + select + rerank kernels */
    uint64_t candidates;   /* candidate pairs that reached the exact re-rank */
    uint64_t fallbacks;    /* queries answered by a dense exact pass (candidate list overflowed twice, or fewer than k candidates) */
    char scan_kernel[64];  /* name of the kernel timed as "scan" */
    uint64_t retries;      /* queries whose candidate list overflowed and that a second filter pass with a tighter threshold answered.
                            * Same layout and history as VecSimGpuStats (VecSim/vec_sim_gpu.h): append-only from here, pinned by tests/test_abi.py */
} vsgpu_stats;
void vsgpu_stats_reset(vsgpu_ctx *ctx);
void vsgpu_stats_get(vsgpu_ctx *ctx, vsgpu_stats *out);
/* poll(user) != 0 stops a top-k call between its launches -- behind the probe + threshold kernels and behind the filter /
 * scan kernel (the stream is drained there first) -- with VSGPU_ERR_TIMEOUT; NULL (default) = no polling, no drains.  The
 * reference polls its timeout callback once per scanned vector (brute_force.h:265); a kernel in flight cannot be recalled,
 * so this is the finest grain the GPU path has. */
void vsgpu_set_poll(vsgpu_ctx *ctx, int (*poll)(void *user), void *user);
/* knobs: "mfma" (0/1: allow the MFMA filter stage), "dense_pairs", "probe_div", "cand_cap" */
int vsgpu_set_option(vsgpu_ctx *ctx, const char *name, long value);

#ifdef __cplusplus
}
#endif
#endif
