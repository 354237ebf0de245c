// iter_kernels.hpp -- device side of the Flat batch iterator (bfs_batch_iterator.h:24-41, bf_batch_iterator.h:61-175).
//
// The reference materialises all n (score, label) pairs once and then, per batch, scans every live pair with a
// bounded heap.  Here the n scores stay in HBM; per batch the GPU finds T = the n_res-th smallest live score
// (three histogram passes over the order-preserving integer image of the float: 11 + 11 + 10 bits, every CU
// busy) and hands back only the live rows with score <= T.  The host replays the reference's heap loop over
// those few rows in the reference's array order -- the same "replay over {score <= T}" argument as the top-K
// scan -- so batches, tie handling included, are what the reference's loop returns.  Returned rows are
// retired by overwriting their score with a quiet NaN (NaN scores are never handed out).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mfma_kernels.hpp"  // float_sort_key

namespace vsg {

constexpr uint32_t ITER_RETIRED = 0x7FC00000u;  // +qNaN

__device__ __forceinline__ bool iter_is_nan(uint32_t bits) { return (bits & 0x7FFFFFFFu) > 0x7F800000u; }

// hist[(key >> shift) & (bins-1)] over the live scores whose key matches `prefix` under `mask`
static __global__ __launch_bounds__(256) void k_iter_hist(const uint32_t *scores, uint32_t n, uint32_t mask, uint32_t prefix, int shift,
                                                   uint32_t bins, uint32_t *hist) {
    __shared__ uint32_t h[2048];
    for (uint32_t i = threadIdx.x; i < bins; i += 256) h[i] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t b = scores[i];
        if (iter_is_nan(b)) continue;
        const uint32_t key = float_sort_key(b);
        if ((key & mask) == prefix) atomicAdd(&h[(key >> shift) & (bins - 1)], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < bins; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// every live row with key <= tkey -> out[{row, score bits}]; count keeps counting past cap (overflow signal)
static __global__ __launch_bounds__(256) void k_iter_compact(const uint32_t *scores, uint32_t n, uint32_t tkey, uint2 *out, uint32_t *count,
                                                      uint32_t cap) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t b = scores[i];
        if (iter_is_nan(b) || float_sort_key(b) > tkey) continue;
        const uint32_t p = atomicAdd(count, 1u);
        if (p < cap) out[p] = make_uint2(i, b);
    }
}

static __global__ __launch_bounds__(256) void k_iter_retire(uint32_t *scores, const uint32_t *rows, uint32_t m) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) scores[rows[i]] = ITER_RETIRED;
}

}  // namespace vsg
