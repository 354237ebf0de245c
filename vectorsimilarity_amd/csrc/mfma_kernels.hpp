// mfma_kernels.hpp -- the HBM-roofline path for fp32 Flat scans with a wide query batch (gfx950).
//
// Exact reference-order fp32 distances cost 2 VALU ops per (row, query, element): at B = 64 that is
// ~12 ms per 10M x 768 scan, 3x the time HBM needs to stream the rows (SURVEY.md §7 hard part 2).
// So the scan is split:
//
//   stage 1 (this file, k_mfma_filter): stream every row ONCE, compute
//        a = |x|^2 + |q|^2 - 2 <bf16(x), bf16(q)>           (one bf16 MFMA term, fp32 accumulate)
//     and a rigorous bound E >= |a - s_ref| (s_ref = the reference-order fp32 score), then
//        probe mode : per (tile, query) min of a + E   -> thresholds tau_q >= T_q   (k_probe_threshold)
//        filter mode: emit (row, query) when a - E <= tau_q  -> superset of {s_ref <= T_q}
//   stage 2 (exact_kernels.hpp, k_exact_pairs): reference-order exact score of every survivor.
//   host: sequential-heap replay (flat_index.cpp) -> bit-identical reply.
//
// Data movement (the kernel is HBM-bound; MFMA/VALU/LDS all run far below their peaks):
//   * rows go HBM -> LDS by `global_load_lds_dwordx4` (no VGPR round trip), 64 rows x 256 B per stage,
//     whole 256-B row segments per 16 lanes (full-line coalescing), 3-stage ring, counted vmcnt waits
//     and one raw s_barrier per stage so two stages stay in flight per workgroup;
//   * the LDS image is XOR-swizzled on the *source* address (slot ^= row & 15) so the MFMA A-operand
//     reads (ds_read_b128, 16 rows x 32 B per lane group) are bank-conflict free;
//   * the 64 queries live in VGPRs as bf16 B-operands for the whole kernel (16 per wave);
//   * fp32 -> bf16 (RNE) is done in registers between the LDS read and the MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "exact_kernels.hpp"

namespace vsg {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x8_t __attribute__((ext_vector_type(8)));

enum MfmaMode { MF_PROBE = 0, MF_FILTER = 1, MF_STREAM = 2 };
// MF_STREAM (round 6): the filter with a threshold that TIGHTENS while it streams.  tau_q starts as the k-th smallest tile minimum
// of a SMALL probe (a few hundred tiles instead of n / 48 rows) whose k smallest minima also seed a per-query list in global memory;
// every workgroup re-reads tau_q once per tile (an L2 hit, beside the slab pointers it loads there anyway) and, whenever one of its
// tiles' minimum upper bound beats tau_q, enters it into the list (lock-free: mf_stream_insert).  The
// list always holds upper bounds of k DISTINCT rows (tiles are disjoint; the probe's own tiles never insert again), so tau_q >= T_q
// at every moment, and a stale tau_q is only looser: the emitted set stays a superset of {s_ref <= T_q}.  Expected survivors per
// query ~ 2.3 k ln(n / probe rows) whatever the probe's size, which is what lets the probe shrink.
constexpr int MF_KLIST = 128;         // list entries per query (k beyond this: the plain filter)

constexpr int MF_TILE_ROWS = 64;      // default rows per workgroup tile (4 MFMA M-tiles of 16)
constexpr int MF_STAGE_BYTES = 16384; // one ring slot: RT rows x (4096/RT) floats
constexpr int MF_NSTAGE_MAX = 4;
constexpr int MF_QTILE = 64;          // queries per workgroup (16 per wave)
constexpr int MF_NORM_BYTES = 4 * 2 * 256;  // [wave][parity][64 floats]
// Candidate emission goes through a small LDS queue (LDS atomics use lgkmcnt, not the VM counter the
// DMA pipeline is counted on); the queue is flushed to global memory when half full and at kernel end.
constexpr int MF_EQ_CAP = 384;                     // queued {row, query, score bits} records
constexpr int MF_EQ_BYTES = 16 + MF_EQ_CAP * 16;
constexpr int MF_WQ_CAP = MF_EQ_CAP / 4;           // k_mfma_filter: a wave's share of the queue (four waves)
constexpr int mf_lds_bytes(int nstage) { return nstage * MF_STAGE_BYTES + MF_NORM_BYTES + MF_EQ_BYTES; }
// MF_PROBE: the (tile, query) minima wait in LDS and leave in batches (a global store per tile shares the VM counter with
// the ring: one full drain per tile; batching them measured neutral, the probe's rate is set by its short per-workgroup
// runs): 4 waves x MF_PM_TILES tiles x 16 queries
constexpr int MF_PM_TILES = 16;   // (keeps the probe within the default 64 KiB of dynamic LDS)
constexpr int mf_probe_lds_bytes(int nstage) { return mf_lds_bytes(nstage) + 4 * MF_PM_TILES * 64; }

// The queue is touched with inline-asm DS instructions on purpose: for a compiler-visible LDS store or
// atomic hipcc inserts `s_waitcnt vmcnt(0)` while LDS-DMA writes are in flight (it cannot prove the
// addresses disjoint), which would drain the staging ring at every emission.
typedef unsigned int mf_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mf_lds_offset(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}
__device__ __forceinline__ uint32_t mf_queue_reserve(uint32_t eq_n_off) {
    uint32_t ret, one = 1u;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ret) : "v"(eq_n_off), "v"(one) : "memory");
    return ret;
}
__device__ __forceinline__ void mf_queue_write(uint32_t slot_off, uint32_t row, uint32_t q, uint32_t bits) {
    mf_u32x4 v = {row, q, bits, 0u};
    asm volatile("ds_write_b128 %0, %1" ::"v"(slot_off), "v"(v) : "memory");
}

// Barrier that licenses overwriting a ring slot.  A raw s_barrier is not enough: hipcc software-pipelines the
// ds_reads of the unit that just ended across it (their lgkmcnt wait lands after the barrier, next to the MFMAs
// that consume them), so a wave could arrive with reads of the old slot still queued while another wave's DMA
// for the new unit -- fast when it hits L2, as the clamped prefetches past the last tile do -- was already
// rewriting that slot.  Observed as one candidate lost per ~700 query passes (tools/stress_flat.py).
__device__ __forceinline__ void mf_ring_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// all threads of the workgroup: move the queued records to the per-query candidate lists
template <int NTHREADS>
__device__ __forceinline__ void mf_flush_queue(uint32_t *eq_n, uint4 *eq, uint32_t *counts, uint2 *cand, uint32_t cap) {
    const uint32_t n = min(*eq_n, (uint32_t)MF_EQ_CAP);
    for (uint32_t i = threadIdx.x; i < n; i += NTHREADS) {
        const uint4 r = eq[i];
        const uint32_t s = atomicAdd(&counts[r.y], 1u);
        if (s < cap) cand[(size_t)r.y * cap + s] = make_uint2(r.x, r.z);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) *eq_n = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

struct MfmaParams {
    const char *const *slabs;        // row slabs
    const float *const *norm_slabs;  // |x|^2 per row, same slab geometry
    uint32_t slab_shift, slab_mask;
    uint32_t row_stride;             // bytes (dim*4)
    uint32_t n_rows;
    // tiles processed by this launch: tile t (0 <= t < n_tiles) covers rows (tile_first + t*tile_step)*64 ...
    uint32_t tile_first, tile_step, n_tiles;
    // Probe launches sample the table in RUNS of 2^tile_run_shift consecutive tiles, a run every tile_step runs: tile t is
    // table tile tile_first + (t >> s) * (tile_step << s) + (t & (2^s - 1)); s = 0: one tile every tile_step tiles.
    uint32_t tile_run_shift;
    const uint4 *qfrag;              // [q_tile][wave][KSTEPS][lane] bf16x8 B-operand fragments
    const float *qn2;                // [q_tiles*64] |q|^2 (0 for padding queries)
    float cE;                        // E = cE * (|x|^2 + |q|^2) + absE
    float absE;
    int is_l2;                       // 1: a = nx2 + nq2 - 2 dot ; 0: a = 1 - dot
    int iepi;                        // k_mfma_filter_wide on int8 / uint8 rows: 2 = L2, 3 = IP, 4 = Cosine, 5 = uint8 IP (LowpEpi values)
    float *tilemin;                  // MF_PROBE: [q_tiles*64][tilemin_stride]
    size_t tilemin_stride;
    const float *tau;                // MF_FILTER: [q_tiles*64] (-inf for padding queries)
    uint32_t *counts;                // [q_tiles*64]
    uint2 *cand;                     // [q_tiles*64][cap] {row, lower-bound bits}
    uint32_t cap;
    const float *qmeta;              // k_mfma_filter_wide on uint8 Cosine rows (EK = 5): [queries][8] = {norm_q, bits(int 128 sum q' + 16384 dim), ...}
    // MF_STREAM: klist [queries][MF_KLIST] order-preserving keys
    // (a query's slots in cache lines of its own: a slot-major layout -- one 64-byte request for a wave's sixteen thresholds -- put the
    // atomics of all queries on the same lines and cost 0.1 ms more per 10 M rows, r06_stream_ab.txt) of the k smallest upper bounds seen so far, ascending (seeded by
    // k_probe_threshold); the query's threshold is slot k - 1, read again every tile.  k = 0xFFFFFFFF: no insertions (measurement)
    uint32_t *klist;
    uint32_t klist_stride;   // words between two queries' lists (>= MF_KLIST)
    uint32_t k;
    uint32_t refresh_mask, refresh_early;   // re-read tau_q every (mask + 1)-th tile, and after each of a workgroup's first `early` tiles
    uint32_t probe_shift, probe_tiles;   // tiles t = j << probe_shift, j < probe_tiles, were the probe's: their minima are in the list already
};

// One lane, one query: `up` -- an upper bound of a row no list entry stands for -- enters the query's list of the k smallest such bounds.
// LOCK-FREE (the first version took a per-query lock: an unbounded spin hung the GPU -- sixteen lanes of a wave hold sixteen locks --, a
// bounded one cost 0.7 ms per 10 M rows in stalls).  The list is k slots of order-preserving integer keys, ascending; inserting x is the
// cascade  x = max(x, atomicMin(&slot[i], x))  from the first slot whose value exceeds x: every atomic step leaves {slot, carry} the
// same multiset with the smaller one in the slot, so (1) the slots always hold bounds of k DISTINCT rows (what falls off the end is the
// largest), (2) each slot only ever decreases and slot[i] <= slot[i + 1] holds after every single step, whatever interleaves -- hence
// tau_q = slot[k - 1], read by a plain load, is at every moment the k-th smallest bound of k distinct rows: >= T_q.  Slots at or below x
// are skipped without an atomic (they can only have decreased since they were read).
__device__ __forceinline__ uint32_t mf_key_of(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float mf_float_of(uint32_t key) {
    return __uint_as_float(key ^ ((key & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu));
}
// Returns the key of slot k - 1 as last seen: the caller's threshold for free.
__device__ __forceinline__ uint32_t mf_stream_insert(uint32_t *klist, uint32_t stride, uint32_t k, int q, float up) {
    uint32_t *lst = klist + (size_t)q * stride;
    uint32_t x = mf_key_of(up);
    // the slots as they are now, the last one and eight at a time from the first, all loads in flight together
    uint32_t last = __hip_atomic_load(lst + k - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t i = 0;
    for (uint32_t i0 = 0; i0 < k; i0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (i0 + j < k) ? __hip_atomic_load(lst + i0 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
        if (last <= x) return last;   // (the caller's threshold was stale: nothing to enter)
        uint32_t below = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) below += (i0 + j < k && v[j] <= x) ? 1u : 0u;
        i += below;
        if (below < 8) break;
    }
    for (; i < k; i++) {
        const uint32_t old = atomicMin(lst + i, x);
        if (i + 1 == k) last = old < x ? old : x;
        if (old <= x) continue;    // (the slot had come down to x or below meanwhile: x moves on unchanged)
        x = old;                   // x rests in slot i; what it displaced moves on
        if (i + 1 < k) {
            last = __hip_atomic_load(lst + k - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (last <= x) break;  // the carry is the largest: it falls off
        }
    }
    return last;
}

template <int AUX>
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_byte_off, char *lds_base) {
    // 64 lanes x 16 B -> LDS [lds_base + lds_byte_off + lane*16]; lds address is wave-uniform.
    // AUX = 2 marks the stream non-temporal (every row is read exactly once per launch).
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)gsrc,
        (__attribute__((address_space(3))) void *)(lds_base + lds_byte_off), 16, 0, AUX);
}
__device__ __forceinline__ void glds4(const void *gsrc, uint32_t lds_byte_off, char *lds_base) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)gsrc,
        (__attribute__((address_space(3))) void *)(lds_base + lds_byte_off), 4, 0, 0);
}

// RT = rows per workgroup tile: 64 (stage = 64 rows x 256 B, for any dim % 64 == 0) or 16 (stage = 16 rows
// x 1 KiB, dim % 256 == 0: every DMA instruction moves 1 KiB of ONE row, which HBM likes better).
// XOPT (tuning bits): 1 = slab pointers by cached scalar loads, 2 = one norm copy per workgroup (wave 0 requests it),
// 4 = branch-free "any survivor?" pass in front of the emitting loop
// EB = 8: fp64 rows (L2_AVX512F_FP64.h:11-59 / IP twin are the reference kernels; their scores come from the exact re-rank).
// Same ring and swizzle over bytes; a 256-byte window holds 32 doubles = one MFMA k-step, a lane's 8 elements are four
// 16-byte pieces, converted f64 -> f32 -> bf16 in registers.  The bound E covers that rounding like the fp32 one (the
// reference's own accumulation error, in double, is far below the fp32 figure the constant budgets for).
template <int KSTEPS, int MODE, int NS = 3, int AUX = 0, int MINW = 1, int RT = 64, int XOPT = 0, int EB = 4>
__global__ __launch_bounds__(256, MINW) void k_mfma_filter(MfmaParams P) {
    constexpr int MF_NSTAGE = NS;
    constexpr bool SLOAD = (XOPT & 1) != 0, SNORM = (XOPT & 2) != 0, PRESCREEN = (XOPT & 4) != 0;
    static_assert(NS == 3 || NS == 4, "ring depth");
    static_assert(RT == 64 || RT == 16, "tile rows");
    constexpr int MT = RT / 16;                      // MFMA M-tiles per tile
    static_assert(EB == 4 || EB == 8, "fp32 or fp64 rows");
    constexpr int KC = (MF_STAGE_BYTES / EB) / RT;   // elements per row per stage: fp32 64 or 256, fp64 32 or 128
    constexpr int SEG = KC * EB;                     // bytes per row per stage: 256 or 1024
    constexpr int KSUB = KC / 32;                    // MFMA k-steps per stage: fp32 2 or 8, fp64 1 or 4
    static_assert(KSTEPS % KSUB == 0, "dim must be a multiple of the stage width");
    constexpr int KCH = KSTEPS / KSUB;               // stages per row tile
    static_assert(KCH >= NS - 1, "tile shorter than the ring");
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15;   // A: row inside an M-tile / B,C: query inside the wave's 16
    const int kq = lane >> 4;    // A,B: which 8 of the 32 k ; C: row quad
    const int qtile = blockIdx.y;

    // ---- resident query fragments (bf16 B operands) ----
    bf16x8_t qf[KSTEPS];
    {
        const uint4 *src = P.qfrag + ((size_t)(qtile * 4 + wave) * KSTEPS) * 64 + lane;
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) {
            uint4 v = src[(size_t)s * 64];
            qf[s] = __builtin_bit_cast(bf16x8_t, v);
        }
    }
    const int qidx = qtile * MF_QTILE + wave * 16 + m16;
    float nq2 = P.qn2[qidx];
    float tau = 0.f;
    if (MODE != MF_PROBE) tau = P.tau[qidx];
    // Pin the completion of these ordinary loads HERE, before any LDS-DMA is issued: an empty asm that
    // consumes the registers makes hipcc place its s_waitcnt now instead of a vmcnt(0) in front of the
    // first MFMA of every tile (which would drain the stages in flight once per tile).
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) asm volatile("" : "+v"(qf[s]));
    asm volatile("" : "+v"(nq2), "+v"(tau));

    // ---- per-lane staging geometry ----
    // A stage image is 16 KiB = 16 DMA instructions of 1 KiB; this wave issues instructions 4w..4w+3.
    // Instruction g, lane l fills LDS byte L = 1024 g + 16 l: row = L / SEG, 16-B slot = (L % SEG)/16,
    // 256-B window = slot/16, slot-in-window p = slot%16, and its SOURCE chunk is p ^ (row & 15):
    // the XOR swizzle lives on the source address because the DMA destination is lane-linear.
    uint32_t st_row[4];   // row inside the tile
    uint32_t st_off[4];   // byte offset inside the row's stage segment
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t L = 1024u * (uint32_t)(4 * wave + t) + 16u * (uint32_t)lane;
        const uint32_t row = L / SEG, slot = (L % SEG) / 16;
        st_row[t] = row;
        st_off[t] = (slot / 16) * 256 + (((slot % 16) ^ (row & 15)) * 16);
    }
    const uint32_t lds_stage_wave_off = (uint32_t)(wave * 4096);
    char *norm_lds = lds + MF_NSTAGE * MF_STAGE_BYTES + (SNORM ? 0 : wave) * 512;
    const bool norm_loader = !SNORM || wave == 0;
    // survivors wait in a queue of this WAVE's own (MF_WQ_CAP records of the workgroup's MF_EQ_CAP) and leave in batches, wave by wave:
    // its fill count is a register, nobody else touches it, so neither emission nor flush meets a barrier or a returning LDS atomic
    const uint32_t wq_off = mf_lds_offset(lds + MF_NSTAGE * MF_STAGE_BYTES + MF_NORM_BYTES + 16) + (uint32_t)wave * (uint32_t)(MF_WQ_CAP * 16);
    uint32_t wq_n = 0;
    auto flush_wave_queue = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the queued records have landed
        const uint32_t n = min(wq_n, (uint32_t)MF_WQ_CAP);
        for (uint32_t i = (uint32_t)lane; i < n; i += 64) {
            mf_u32x4 r;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(wq_off + i * 16) : "memory");
            const uint32_t s = atomicAdd(&P.counts[r.y], 1u);
            if (s < P.cap) P.cand[(size_t)r.y * P.cap + s] = make_uint2(r.x, r.z);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (stores / atomics share the VM counter with the staged loads)
        wq_n = 0;
    };

    const uint32_t my_first = blockIdx.x;
    const uint32_t step = gridDim.x;

    // (round 4: a "blocked" probe -- a workgroup's j-th sample tile next to its (j - 1)-th instead of gridDim.x sample positions
    // further, so that its slab pointers stay put -- changed no batch time: profiles/r04_probe_blocked.txt; removed)
    auto tile_row0 = [&](uint32_t t) -> uint32_t {
        return (P.tile_first + (t >> P.tile_run_shift) * (P.tile_step << P.tile_run_shift) + (t & ((1u << P.tile_run_shift) - 1u))) * RT;
    };
    const char *rp_cur[4], *rp_nxt[4];
    const float *np_cur, *np_nxt;
    uint32_t cur_slab = 0xFFFFFFFFu;          // SLOAD: slab pointers cached in SGPRs, inline-asm scalar loads
    uint64_t cur_sbase = 0, cur_nbase = 0;
    auto make_ptrs = [&](uint32_t t, const char *(&rp)[4], const float *&np) {
        uint32_t tt = t < P.n_tiles ? t : P.n_tiles - 1;  // past-the-end prefetches re-read the last tile
        const uint32_t r0 = tile_row0(tt);
        // a tile never straddles a slab (slab rows are a power of two >= 64), so the slab lookups are wave-uniform.
        // hipcc turns them into two vector loads per tile whose vmcnt(0) drains this workgroup's ring once per tile;
        // replacing them with cached inline-asm scalar loads (as k_mfma_filter_lowp does) was measured SLOWER here
        // (4.65 vs 4.33 ms on 10 M x 768: with 2-3 workgroups per CU the drain seems to pace them usefully)
        const uint32_t sidx = __builtin_amdgcn_readfirstlane(r0 >> P.slab_shift);
        const char *sbase;
        const float *nbase;
        if (SLOAD) {
            if (sidx != cur_slab) {
                cur_slab = sidx;
                const char *const *sp = P.slabs + sidx;
                const float *const *npp = P.norm_slabs + sidx;
                asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                             : "=&s"(cur_sbase), "=&s"(cur_nbase)
                             : "s"(sp), "s"(npp)
                             : "memory");
            }
            sbase = reinterpret_cast<const char *>(cur_sbase);
            nbase = reinterpret_cast<const float *>(cur_nbase);
        } else {
            sbase = P.slabs[sidx];
            nbase = P.norm_slabs[sidx];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t row = r0 + st_row[i];
            if (row >= P.n_rows) row = P.n_rows - 1;  // tails read valid memory; masked in the epilogue
            rp[i] = sbase + (size_t)(row & P.slab_mask) * P.row_stride + st_off[i];
        }
        uint32_t nrow = r0 + lane;
        if (nrow >= P.n_rows) nrow = P.n_rows - 1;
        np = nbase + (nrow & P.slab_mask);
    };

    // issue the loads of one stage: k-chunk `kc` of the tile whose pointers are given
    auto issue = [&](const char *const (&rp)[4], const float *np, int kc, uint32_t slot, bool with_norm,
                     uint32_t norm_parity) {
        const uint32_t base = slot * MF_STAGE_BYTES + lds_stage_wave_off;
#pragma unroll
        for (int i = 0; i < 4; i++) glds16<AUX>(rp[i] + (size_t)kc * SEG, base + i * 1024, lds);
        if (with_norm && norm_loader) glds4(np, norm_parity * 256, norm_lds);
    };

    uint32_t tile = my_first;
    uint32_t tiles_done = 0;   // (MF_STREAM: paces the threshold re-reads)
    make_ptrs(tile, rp_cur, np_cur);
    make_ptrs(tile + step, rp_nxt, np_nxt);
    uint32_t slot_c = 0;       // slot computed in the current unit
    uint32_t parity = 0;       // norm ring parity of the current tile

    // prologue: the first NS-1 units.  Unit index u (0-based from the first tile) is (tile + u/KCH, u%KCH).
#pragma unroll
    for (int u = 0; u < NS - 1; u++) {
        if (u < KCH) issue(rp_cur, np_cur, u, u, u == 0, 0);
        else issue(rp_nxt, np_nxt, u - KCH, u, u == KCH, 1);
    }

    // MF_STREAM: this wave's threshold mailbox behind the queue, a key per lane (the four lanes of a query hold the same)
    const uint32_t tau_box = mf_lds_offset(lds) + (uint32_t)mf_lds_bytes(MF_NSTAGE) + (uint32_t)wave * 256u + (uint32_t)lane * 4u;
    if (MODE == MF_STREAM) asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(tau_box), "v"(mf_key_of(tau)) : "memory");
    // MF_PROBE: this wave's buffered tile minima, [MF_PM_TILES][16 queries] floats behind the queue
    const uint32_t pm_off = mf_lds_offset(lds) + (uint32_t)mf_lds_bytes(MF_NSTAGE) + (uint32_t)wave * (MF_PM_TILES * 64) + (uint32_t)m16 * 4u;
    uint32_t pm_n = 0, pm_tile0 = 0;
    auto flush_probe_minima = [&]() {   // lane (kq, m16) writes out the tiles kq, kq + 4, ... of query column m16
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (uint32_t it = (uint32_t)kq; it < pm_n; it += 4) {
            float v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(pm_off + it * 64u) : "memory");
            P.tilemin[(size_t)qidx * P.tilemin_stride + pm_tile0 + it * step] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pm_n = 0;
    };
    for (; tile < P.n_tiles; tile += step) {
        f32x4_t acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int c = 0; c < KCH; c++) {
            // unit (tile,c) must have landed; the NS-2 younger units (4 loads each, +1 norm load for a
            // unit that opens a tile) may stay in flight
            {
                constexpr int AHEAD = NS - 2;
                int allowed = 4 * AHEAD;
#pragma unroll
                for (int a = 1; a <= AHEAD; a++)
                    if ((c + a) % KCH == 0) allowed += 1;
                if (!norm_loader) allowed = 4 * AHEAD;
                if (allowed == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (allowed == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else if (allowed == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (allowed == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            }
            mf_ring_barrier();
            // refill the slot read in the previous unit with unit +(NS-1)
            {
                constexpr int D = NS - 1;
                uint32_t slot_p = slot_c + D;
                if (slot_p >= MF_NSTAGE) slot_p -= MF_NSTAGE;
                const int cc = c + D;
                if (cc < KCH) issue(rp_cur, np_cur, cc, slot_p, false, 0);
                else if (cc < 2 * KCH) issue(rp_nxt, np_nxt, cc - KCH, slot_p, cc == KCH, parity ^ 1u);
                else {
                    // only when KCH < NS-1 would a unit two tiles ahead be needed; excluded by static_assert
                }
            }
            const char *sbase = lds + slot_c * MF_STAGE_BYTES;
#pragma unroll
            for (int j = 0; j < KSUB; j++) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    f32x8_t x;
                    if constexpr (EB == 4) {
                        const char *rowp = sbase + (mt * 16 + m16) * SEG + (j / 2) * 256;
                        const int p0 = (8 * (j % 2) + 2 * kq) ^ m16;
                        const int p1 = (8 * (j % 2) + 2 * kq + 1) ^ m16;
                        f32x4_t lo = *reinterpret_cast<const f32x4_t *>(rowp + p0 * 16);
                        f32x4_t hi = *reinterpret_cast<const f32x4_t *>(rowp + p1 * 16);
                        x = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    } else {
                        typedef double f64x2_t __attribute__((ext_vector_type(2)));
                        const char *rowp = sbase + (mt * 16 + m16) * SEG + j * 256;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const f64x2_t d = *reinterpret_cast<const f64x2_t *>(rowp + (((4 * kq + i) ^ m16) * 16));
                            x[2 * i] = (float)d[0];
                            x[2 * i + 1] = (float)d[1];
                        }
                    }
                    bf16x8_t a = __builtin_convertvector(x, bf16x8_t);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[c * KSUB + j], acc[mt], 0, 0, 0);
                }
            }
            slot_c = slot_c + 1 == MF_NSTAGE ? 0 : slot_c + 1;
        }

        // ---- epilogue: lane holds dot(row = mt*16 + kq*4 + i, query = m16) ----
        const uint32_t r0 = tile_row0(tile);
        const float *nrm = reinterpret_cast<const float *>(norm_lds + parity * 256);
        bool emitted = false;
        float tmin = INFINITY;
        bool skip = false;
        if (PRESCREEN && MODE != MF_PROBE) {
            bool any = false;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                f32x4_t n4 = *reinterpret_cast<const f32x4_t *>(nrm + mt * 16 + kq * 4);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float ssum = n4[i] + nq2;
                    const float a = P.is_l2 ? (ssum - 2.0f * acc[mt][i]) : (1.0f - acc[mt][i]);
                    any |= !(a - (P.cE * ssum + P.absE) > tau);
                }
            }
            skip = !__any(any);
        }
        if (!skip)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            f32x4_t n4 = *reinterpret_cast<const f32x4_t *>(nrm + mt * 16 + kq * 4);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t row = r0 + mt * 16 + kq * 4 + i;
                const float nx2 = n4[i];
                const float ssum = nx2 + nq2;
                const float dot = acc[mt][i];
                const float a = P.is_l2 ? (ssum - 2.0f * dot) : (1.0f - dot);
                const float E = P.cE * ssum + P.absE;
                if (MODE == MF_STREAM) {
                    const float up = a + E;
                    if (row < P.n_rows && up < tmin) tmin = up;
                }
                if (MODE == MF_PROBE) {
                    const float up = a + E;
                    if (row < P.n_rows && up < tmin) tmin = up;
                } else {
                    const float low = a - E;
                    const bool pass = row < P.n_rows && !(low > tau);  // NaN bounds (NaN/Inf in the data) go on to the exact re-rank
                    // the wave's own queue: a slot is this lane's rank among the passing lanes behind the wave's count -- a ballot and a
                    // fire-and-forget LDS write, no returning LDS atomic to wait for (round 6: the shared queue's ds_add_rtn + wait cost
                    // a workgroup 0.3-0.6 us per survivor, profiles/r06_wave_queue.txt)
                    const uint64_t pm = __ballot(pass);
                    if (pm) {
                        const uint32_t pos = wq_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                        if (pass) {
                            if (pos < (uint32_t)MF_WQ_CAP) {
                                mf_queue_write(wq_off + pos * 16, row, (uint32_t)qidx, __float_as_uint(low));
                            } else {  // queue full (dense survivors): straight to global memory
                                uint32_t s = atomicAdd(&P.counts[qidx], 1u);
                                if (s < P.cap) P.cand[(size_t)qidx * P.cap + s] = make_uint2(row, __float_as_uint(low));
                                emitted = true;
                            }
                        }
                        wq_n += (uint32_t)__popcll(pm);
                    }
                }
            }
        }
        if (MODE == MF_PROBE) {
            // min over the 4 row quads (lanes m16, m16+16, m16+32, m16+48)
            tmin = fminf(tmin, __shfl_xor(tmin, 16));
            tmin = fminf(tmin, __shfl_xor(tmin, 32));
            if (kq == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(pm_off + pm_n * 64u), "v"(tmin) : "memory");
            if (pm_n == 0) pm_tile0 = tile;
            if (++pm_n == (uint32_t)MF_PM_TILES) flush_probe_minima();
        } else {
            if (MODE == MF_STREAM) {
                // this tile's minimum upper bound per query (lanes m16, m16+16, m16+32, m16+48 hold the four row quads)
                tmin = fminf(tmin, __shfl_xor(tmin, 16));
                tmin = fminf(tmin, __shfl_xor(tmin, 32));
                const uint32_t tt = tile - P.tile_first;
                const bool probed = (tt & ((1u << P.probe_shift) - 1u)) == 0 && (tt >> P.probe_shift) < P.probe_tiles;   // (wave-uniform)
                // (the insertion waits for its own loads and atomics -- nothing of it is in flight afterwards)
                {   // the mailbox: the threshold's key as of the last re-read
                    uint32_t kraw;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(kraw) : "v"(tau_box) : "memory");
                    tau = fminf(tau, mf_float_of(kraw));
                }
                uint32_t seen = 0xFFFFFFFFu;
                if (kq == 0 && !probed && tmin < tau && P.k != 0xFFFFFFFFu) seen = mf_stream_insert(P.klist, P.klist_stride, P.k, qidx, tmin);
                if (__any(seen != 0xFFFFFFFFu)) {   // the four lanes of a query share what the insertion saw of its threshold
                    const uint32_t s0 = (uint32_t)__shfl((int)seen, m16);
                    if (s0 != 0xFFFFFFFFu) tau = fminf(tau, mf_float_of(s0));
                }
            }
            if (__any(emitted)) {
                // stores/atomics share the VM counter with the staged loads: drain once so the counted
                // waits of the next tile see only loads
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (wq_n >= (uint32_t)MF_WQ_CAP / 2) flush_wave_queue();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // queue writes land before the next barrier
        }

        // rotate tile state
#pragma unroll
        for (int i = 0; i < 4; i++) rp_cur[i] = rp_nxt[i];
        np_cur = np_nxt;
        make_ptrs(tile + 2 * step, rp_nxt, np_nxt);
        // MF_STREAM: the threshold as it stands now (an L2 hit issued beside the slab-pointer loads; used in the next tile's epilogue)
        // -- every (refresh_mask + 1)-th tile only: a device-scope load is served behind the XCD's own L2 (sc1), and 4096 waves asking
        // for 16 thresholds each every tile were 1.2 TB/s of requests beside the table's stream (the scan 11 % slower: r06_stream_ab.txt)
        if (MODE == MF_STREAM && P.k != 0xFFFFFFFFu && ((++tiles_done & P.refresh_mask) == 0 || tiles_done <= P.refresh_early)) {
            // -- and through the LDS DMA, into this wave's mailbox (lane-linear: lane l's dword): no register waits for it, so no
            // s_waitcnt vmcnt(0) of the compiler's drains the ring for it (a plain load here stalled the workgroup a round trip per
            // re-read); it is one more load in flight, which makes the counted waits stricter, never laxer, and it has landed at the
            // latest when the next tile's slab pointers have
            // (sc1: past the L1, served by the L2 while the line is there; `nt` / `sc1 nt` re-reads were measured 0.7-2.1 ms SLOWER per
            // 10 M rows -- the hint lets the line go, every re-read then comes from memory: r06_stream_ab.txt)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(P.klist + (size_t)qidx * P.klist_stride + (P.k - 1)),
                                             (__attribute__((address_space(3))) void *)(lds + mf_lds_bytes(MF_NSTAGE) + wave * 256), 4, 0, 16 /* sc1 */);
        }
        parity ^= 1u;
    }
    if (MODE == MF_PROBE && pm_n) flush_probe_minima();
    // drain the stages still in flight before the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (MODE != MF_PROBE) flush_wave_queue();
}

// |x|^2 per row in double, rounded once to float (relative error <= 2^-24): feeds the bound E
static __global__ __launch_bounds__(256) void k_row_norms_f32(const char *rows, uint32_t row_stride, uint32_t dim,
                                                       uint32_t n, float *out) {
    const int lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *p = reinterpret_cast<const float *>(rows + (size_t)row * row_stride);
    double s = 0.0;
    for (uint32_t i = lane; i < dim; i += 64) {
        double v = (double)p[i];
        s += v * v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[row] = (float)s;
}

static __global__ __launch_bounds__(256) void k_row_norms_f64(const char *rows, uint32_t row_stride, uint32_t dim,
                                                       uint32_t n, float *out) {
    const int lane = threadIdx.x & 63;
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const double *p = reinterpret_cast<const double *>(rows + (size_t)row * row_stride);
    double s = 0.0;
    for (uint32_t i = lane; i < dim; i += 64) s += p[i] * p[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[row] = (float)s;   // (+inf beyond the float range: the bound turns NaN and the row goes to the re-rank)
}

// ---- stage 3: per-query selection among the exactly re-scored survivors ----
// One workgroup per query: T = k-th smallest exact score (bitwise search over the order-preserving
// integer image of the float), then every candidate with score <= T is compacted to out[q][...].
// The host only sorts those few by id and replays the sequential heap.
__device__ __forceinline__ uint32_t float_sort_key(uint32_t bits) {
    // unsigned order == float order; a NaN of either sign sorts after everything (a negative NaN -- 1 - NaN, 0 / 0 --
    // would otherwise be the smallest key and become the k-th score nothing compares below)
    if ((bits & 0x7FFFFFFFu) > 0x7F800000u) return 0xFFFFFFFFu;
    return bits ^ ((bits & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
}
constexpr uint32_t SEL_LDS_KEYS = 8192;   // candidate keys kept in LDS for the counting passes (the rest is re-read from L2)
// one 256-thread workgroup selects query q: `keys` = KEYS words of LDS, nqs = number of queries (the raw counts sit behind the selected ones)
// The k-th smallest key by radix: four passes of eight bits, each a 256-bin histogram of the keys that share the prefix found so
// far (LDS atomics; a wave first folds the lanes that hit its two most popular bins -- scores of one batch share their exponent,
// so the first pass would otherwise serialise on one address), a scan of the bins and the bin that holds rank kk.  (Rounds 1-3
// found it bit by bit: 32 passes with a barrier and a reduction each, 16-18 us per batch whatever its size -- a quarter of a
// single query's 77 us, profiles/r04_c1_timeline.txt.)
template <uint32_t KEYS>
__device__ __forceinline__ void select_upto_kth_body(uint32_t *keys, uint32_t *hist, uint32_t *sh, uint32_t *wpos, const uint2 *cand, const uint32_t *counts,
                                                     uint32_t cap, uint32_t k, uint2 *out, uint32_t *out_counts, uint32_t out_cap, int q, uint32_t nqs) {
    const uint32_t raw = counts[q];
    if (threadIdx.x == 0) out_counts[nqs + q] = raw;   // the raw count rides along (statistics, "fewer than k" check)
    // (a list that overflowed is selected from all the same: the k-th smallest exact score of the slots that were filled bounds
    // the true k-th score from above, and the host runs one more filter pass with it -- collect_candidates; it knows from the
    // raw count that the list is truncated)
    const uint32_t n = min(raw, cap);
    const uint2 *c = cand + (size_t)q * cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t T = 0xFFFFFFFFu;
    if (n > k) {
        const uint32_t nl = min(n, KEYS);
        for (uint32_t i = threadIdx.x; i < nl; i += 256) keys[i] = float_sort_key(c[i].y);
        uint32_t prefix = 0, kk = k;   // the key sought is the kk-th smallest of those whose bits above `shift + 8` equal `prefix`
        for (int pass = 0; pass < 4 && k > 0; pass++) {
            const int shift = 24 - 8 * pass;
            const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
            hist[threadIdx.x] = 0;
            __syncthreads();   // (also: the keys are staged)
            auto count_key = [&](uint32_t key, bool valid) {
                bool pend = valid && (key & hi_mask) == prefix;
                const uint32_t bin = (key >> shift) & 255u;
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const unsigned long long m = __ballot(pend);
                    if (m == 0) break;
                    const int leader = __ffsll((long long)m) - 1;
                    const uint32_t lb = (uint32_t)__shfl((int)bin, leader);
                    const unsigned long long same = __ballot(pend && bin == lb);
                    if (lane == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
                    pend = pend && bin != lb;
                }
                if (pend) atomicAdd(&hist[bin], 1u);
            };
            for (uint32_t i0 = 0; i0 < nl; i0 += 256) {
                const uint32_t i = i0 + threadIdx.x;
                count_key(i < nl ? keys[i] : 0u, i < nl);
            }
            for (uint32_t i0 = nl; i0 < n; i0 += 256) {
                const uint32_t i = i0 + threadIdx.x;
                count_key(i < n ? float_sort_key(c[i].y) : 0u, i < n);
            }
            __syncthreads();
            const uint32_t v = hist[threadIdx.x];
            uint32_t incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
                if (lane >= o) incl += t;
            }
            if (lane == 63) sh[wave] = incl;
            __syncthreads();
            for (int w = 0; w < wave; w++) incl += sh[w];
            if (incl >= kk && incl - v < kk) {   // exactly one bin: at least kk keys share the prefix
                sh[4] = threadIdx.x;
                sh[5] = kk - (incl - v);
            }
            __syncthreads();
            prefix |= sh[4] << shift;
            kk = sh[5];
        }
        T = prefix;
    }
    if (threadIdx.x == 0) *wpos = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint2 r = c[i];
        const uint32_t bits = r.y;
        const bool is_nan = (bits & 0x7FFFFFFFu) > 0x7F800000u;
        if (!is_nan && float_sort_key(bits) <= T) {
            const uint32_t p = atomicAdd(wpos, 1u);
            if (p < out_cap) out[(size_t)q * out_cap + p] = r;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out_counts[q] = *wpos > out_cap ? 0xFFFFFFFFu : *wpos;
}
static __global__ __launch_bounds__(256) void k_select_upto_kth(const uint2 *cand, const uint32_t *counts, uint32_t cap,
                                                         uint32_t k, uint2 *out, uint32_t *out_counts,
                                                         uint32_t out_cap) {
    __shared__ uint32_t keys[SEL_LDS_KEYS];
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh[8];
    __shared__ uint32_t wpos;
    select_upto_kth_body<SEL_LDS_KEYS>(keys, hist, sh, &wpos, cand, counts, cap, k, out, out_counts, out_cap, (int)blockIdx.x, gridDim.x);
}


// ---- stage 2: exact reference-order scores of the surviving (row, query) pairs ----
// One VL-lane group per pair; pairs of query q are cand[q][0 .. min(counts[q], cap)).
template <int EK, int OPK>
__global__ __launch_bounds__(256) void k_exact_pairs(ScanParams P) {
    using E = Elem<EK>;
    using acc_t = typename E::acc_t;
    constexpr int VL = E::VL;
    constexpr int GROUPS = 256 / VL;
    const int lane = threadIdx.x % VL;
    const int grp = threadIdx.x / VL;
    const int q = blockIdx.y;
    const uint32_t cnt = min(P.counts[q], P.cap);
    const acc_t *qv = reinterpret_cast<const acc_t *>(P.qperm) + (size_t)q * P.steps * VL;
    for (uint32_t s = blockIdx.x * GROUPS + grp; s < cnt; s += gridDim.x * GROUPS) {
        uint2 rec = P.cand[(size_t)q * P.cap + s];
        const uint32_t row = rec.x;
        const char *rp = P.slabs[row >> P.slab_shift] + (size_t)(row & P.slab_mask) * P.row_stride;
        acc_t acc = (acc_t)0;
        // branch-free chunks: CH row loads in flight before the first FMA (idle entries keep the accumulator)
        constexpr int CH = 12;
        for (int s0 = 0; s0 < P.steps; s0 += CH) {
            int off[CH];
            acc_t xv[CH], qq[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) off[j] = (s0 + j < P.steps) ? P.offs[(s0 + j) * VL + lane] : -1;
#pragma unroll
            for (int j = 0; j < CH; j++) xv[j] = E::load(rp + (off[j] >= 0 ? off[j] : 0));
#pragma unroll
            for (int j = 0; j < CH; j++) qq[j] = qv[min(s0 + j, P.steps - 1) * VL + lane];
#pragma unroll
            for (int j = 0; j < CH; j++) {
                const acc_t t = acc_step<OPK>(xv[j], qq[j], acc);
                acc = off[j] >= 0 ? t : acc;
            }
        }
        const typename Reduced<acc_t>::type tot = lane_reduce<VL>((typename Reduced<acc_t>::type)acc, P.reduce);
        if (lane == 0) {
            if constexpr (EK == EK_F64) {   // double scores live beside the candidate list: P.out[q][slot]
                reinterpret_cast<double *>(P.out)[(size_t)q * P.cap + s] = epilogue_score<double>(tot, P.epilogue, 0.f, 0.f);
            } else {
                float sc;
                if constexpr (EK == EK_SQ8 || EK == EK_SQ8H) sc = sq8_score(tot, P.epilogue, P.sq8_fused, rp + P.norm_off, P.qnorm[2 * q], P.qnorm[2 * q + 1]);
                else sc = epilogue_score<float>(tot, P.epilogue, 0.f, 0.f);
                rec.y = __float_as_uint(sc);
                P.cand[(size_t)q * P.cap + s] = rec;
            }
        }
    }
}


// Dense variant: the scores of ALL n rows of one query are in `dense` (column = row id); the k-th smallest key by the same radix
// selection (four passes of eight bits over the n keys, re-read from L2 / HBM: n may be the whole table -- the fallback of a query
// whose candidates overflowed twice, collect_candidates); the survivors (score <= T_k) are compacted as {row, score} records.
// (Rounds 1-3: 32 bitwise counting passes over the n keys.)
// SLICED (round 6): blockIdx.y names a slice of slice_len consecutive rows; the workgroup selects within ITS slice (the k-th smallest of
// the slice bounds the query's k-th score from above, so the rows at or below it in every slice are a superset of the rows at or below
// the true k-th) and appends its survivors to the query's candidate list through a global counter (out_counts[q], zeroed by the caller);
// k_select_upto_kth then selects among the few hundred survivors.  One workgroup per query walked all n keys four times: 40 us at
// n = 100 K, 0.6 ms at 400 K -- which is what kept single queries on small tables on the six-kernel filter path.
template <bool SLICED>
__device__ __forceinline__ void select_dense_body(const float *dense, size_t stride, uint32_t n_all, uint32_t k_all, uint2 *out,
                                                  uint32_t *out_counts, uint32_t out_cap, uint32_t slice_len) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh[8];
    __shared__ uint32_t wpos;
    const int q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t base = SLICED ? blockIdx.y * slice_len : 0u;
    if (SLICED && base >= n_all) return;
    const uint32_t n = SLICED ? min(slice_len, n_all - base) : n_all;
    const uint32_t k = min(k_all, n);
    const uint32_t *c = reinterpret_cast<const uint32_t *>(dense + (size_t)q * stride) + base;
    uint32_t T = 0xFFFFFFFFu;
    if (n > k && k > 0) {
        uint32_t prefix = 0, kk = k;
        for (int pass = 0; pass < 4; pass++) {
            const int shift = 24 - 8 * pass;
            const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
            if (threadIdx.x < 256) hist[threadIdx.x] = 0;
            __syncthreads();
            for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
                const uint32_t i = i0 + threadIdx.x;
                const uint32_t key = i < n ? float_sort_key(c[i]) : 0u;
                bool pend = i < n && (key & hi_mask) == prefix;
                const uint32_t bin = (key >> shift) & 255u;
#pragma unroll
                for (int r = 0; r < 2; r++) {   // the lanes that hit the wave's two most popular bins count as one add each
                    const unsigned long long m = __ballot(pend);
                    if (m == 0) break;
                    const int leader = __ffsll((long long)m) - 1;
                    const uint32_t lb = (uint32_t)__shfl((int)bin, leader);
                    const unsigned long long same = __ballot(pend && bin == lb);
                    if (lane == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
                    pend = pend && bin != lb;
                }
                if (pend) atomicAdd(&hist[bin], 1u);
            }
            __syncthreads();
            uint32_t v = 0, incl = 0;
            if (threadIdx.x < 256) {
                v = hist[threadIdx.x];
                incl = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
                    if (lane >= o) incl += t;
                }
                if (lane == 63) sh[wave] = incl;
            }
            __syncthreads();
            if (threadIdx.x < 256) {
                for (int w = 0; w < wave; w++) incl += sh[w];
                if (incl >= kk && incl - v < kk) {   // exactly one bin: at least kk keys share the prefix
                    sh[4] = threadIdx.x;
                    sh[5] = kk - (incl - v);
                }
            }
            __syncthreads();
            prefix |= sh[4] << shift;
            kk = sh[5];
        }
        T = prefix;
    } else if (k == 0) {
        T = 0;
    }
    if (threadIdx.x == 0) wpos = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const uint32_t bits = c[i];
        const bool is_nan = (bits & 0x7FFFFFFFu) > 0x7F800000u;
        if (!is_nan && float_sort_key(bits) <= T) {
            const uint32_t p = SLICED ? atomicAdd(&out_counts[q], 1u) : atomicAdd(&wpos, 1u);
            if (p < out_cap) out[(size_t)q * out_cap + p] = make_uint2(base + i, bits);
        }
    }
    if (SLICED) return;   // (the raw count IS the list's counter: a list that ran over is seen by the next kernel and by the host)
    __syncthreads();
    if (threadIdx.x == 0) out_counts[q] = wpos > out_cap ? 0xFFFFFFFFu : wpos;
}
static __global__ __launch_bounds__(1024) void k_select_dense_upto_kth(const float *dense, size_t stride, uint32_t n, uint32_t k,
                                                                uint2 *out, uint32_t *out_counts, uint32_t out_cap) {
    select_dense_body<false>(dense, stride, n, k, out, out_counts, out_cap, 0u);
}
static __global__ __launch_bounds__(1024) void k_select_dense_slices(const float *dense, size_t stride, uint32_t n, uint32_t k, uint2 *out,
                                                              uint32_t *out_counts, uint32_t out_cap, uint32_t slice_len) {
    select_dense_body<true>(dense, stride, n, k, out, out_counts, out_cap, slice_len);
}

// fp64 twin: scores are doubles, keys the order-preserving 64-bit image, records {row, score bits}
struct SelRec64 {
    unsigned long long row;
    unsigned long long bits;
};
__device__ __forceinline__ unsigned long long double_sort_key(unsigned long long b) {
    if ((b & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull) return ~0ull;  // NaN: after everything, see float_sort_key
    return b ^ ((b >> 63) ? ~0ull : 0x8000000000000000ull);
}
// candidate-list twin of k_select_upto_kth for fp64 tables: rows in cand[q][i].x, exact double scores in sc[q][i]
constexpr uint32_t SEL64_LDS_KEYS = 4096;
static __global__ __launch_bounds__(256) void k_select_upto_kth_f64(const uint2 *cand, const double *sc, const uint32_t *counts,
                                                             uint32_t cap, uint32_t k, SelRec64 *out, uint32_t *out_counts,
                                                             uint32_t out_cap) {
    __shared__ unsigned long long keys[SEL64_LDS_KEYS];
    __shared__ uint32_t red[2][4];
    __shared__ uint32_t wpos;
    const int q = blockIdx.x;
    const uint32_t raw = counts[q];
    if (threadIdx.x == 0) out_counts[gridDim.x + q] = raw;
    if (raw > cap) {
        if (threadIdx.x == 0) out_counts[q] = 0xFFFFFFFFu;
        return;
    }
    const uint32_t n = raw;
    const uint2 *c = cand + (size_t)q * cap;
    const unsigned long long *v = reinterpret_cast<const unsigned long long *>(sc + (size_t)q * cap);
    unsigned long long T = ~0ull;
    if (n > k) {
        const uint32_t nl = min(n, SEL64_LDS_KEYS);
        for (uint32_t i = threadIdx.x; i < nl; i += 256) keys[i] = double_sort_key(v[i]);
        __syncthreads();
        T = 0;
        for (int bit = 63; bit >= 0; bit--) {
            const unsigned long long trial = T | (1ull << bit);
            uint32_t cnt = 0;
            for (uint32_t i = threadIdx.x; i < nl; i += 256) cnt += (keys[i] < trial) ? 1u : 0u;
            for (uint32_t i = nl + threadIdx.x; i < n; i += 256) cnt += (double_sort_key(v[i]) < trial) ? 1u : 0u;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
            uint32_t *r = red[bit & 1];
            if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = cnt;
            __syncthreads();
            const uint32_t total = r[0] + r[1] + r[2] + r[3];
            if (total < k) T = trial;
        }
    }
    if (threadIdx.x == 0) wpos = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const unsigned long long bits = v[i];
        const bool is_nan = (bits & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull;
        if (!is_nan && double_sort_key(bits) <= T) {
            const uint32_t p = atomicAdd(&wpos, 1u);
            if (p < out_cap) out[(size_t)q * out_cap + p] = SelRec64{c[i].x, bits};
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out_counts[q] = wpos > out_cap ? 0xFFFFFFFFu : wpos;
}

static __global__ __launch_bounds__(1024) void k_select_dense_upto_kth_f64(const double *dense, size_t stride, uint32_t n, uint32_t k,
                                                                    SelRec64 *out, uint32_t *out_counts, uint32_t out_cap) {
    __shared__ uint32_t red[16];
    __shared__ uint32_t wpos;
    const int q = blockIdx.x;
    const unsigned long long *c = reinterpret_cast<const unsigned long long *>(dense + (size_t)q * stride);
    unsigned long long T = ~0ull;
    if (n > k) {
        T = 0;
        for (int bit = 63; bit >= 0; bit--) {
            const unsigned long long trial = T | (1ull << bit);
            uint32_t cnt = 0;
#pragma unroll 4
            for (uint32_t i = threadIdx.x; i < n; i += 1024) cnt += (double_sort_key(c[i]) < trial) ? 1u : 0u;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
            __syncthreads();
            uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < 16; w++) total += red[w];
            if (total < k) T = trial;
        }
    }
    if (threadIdx.x == 0) wpos = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024) {
        const unsigned long long bits = c[i];
        const bool is_nan = (bits & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull;
        if (!is_nan && double_sort_key(bits) <= T) {
            const uint32_t p = atomicAdd(&wpos, 1u);
            if (p < out_cap) out[(size_t)q * out_cap + p] = SelRec64{i, bits};
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out_counts[q] = wpos > out_cap ? 0xFFFFFFFFu : wpos;
}

}  // namespace vsg
