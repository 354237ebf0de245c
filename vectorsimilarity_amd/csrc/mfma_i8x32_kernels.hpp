// mfma_i8x32_kernels.hpp -- int8 / uint8 filter for 1 KiB rows on v_mfma_i32_32x32x32_i8 (BASELINE config 3's shape:
// int8 Cosine, d = 1024, 256 queries per pass).
//
// Why a second int8 kernel (k_mfma_filter_lowp<LP_I8> stays for the other widths, the probe and 16-byte aux records):
// config 3 is MFMA-bound (26.2 POP per batch against 51.4 GB), and the 16 x 16 x 64 kernel runs its phases back to back --
// 16 lock-step waves request the refill, read fragments + multiply, then screen, and the matrix pipe idles through the
// first and the last (profiles/r02_i8_ksplit.txt).  This kernel is built so that every wave is ONE continuous MFMA stream:
//   * v_mfma_i32_32x32x32_i8, 8 waves x 32 queries: one 16-byte A fragment (32 rows x 32 bytes of k) feeds 64 K MACs, half
//     the LDS fragment traffic and half the MFMA issues per MAC of the 16 x 16 x 64 shape; the 32 query fragments of a wave
//     (128 registers) sit in AGPRs, two waves per SIMD;
//   * the screening of unit u-1 (accumulators double-buffered) and the LDS-DMA requests for unit u+3 are interleaved INTO the
//     MFMA stream of unit u (sched_group_barrier), so nothing but the ring barrier separates two units' MFMAs;
//   * EARLY: a unit's rows are required to have landed one barrier early, so the first fragments of unit u+1 are read before
//     the barrier that ends unit u and the stream restarts without an LDS round trip.
// Results are exact integers (the reference's own epilogues, L2.cpp:164-174, IP.cpp:258-286, applied to the int32 dot), as in
// the 16 x 16 x 64 kernel: E = 0, no re-rank.  C/D layout of the 32 x 32 shape: lane l holds query (l & 31) and the 16 rows
// (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).  A / B operands: lane l holds row / query (l & 31), bytes 16 (l >> 5) .. +16 of the
// 32-byte k-step (the same chunk on both sides is all the dot product needs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mfma_lowp_kernels.hpp"

namespace vsg {

typedef int i32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2l_t __attribute__((ext_vector_type(2)));

constexpr int X32_RT = 32;            // rows per unit (= per tile)
constexpr int X32_ROWB = 1040;        // LDS bytes per row image: 1 KiB + 16, so that 32 rows x one 16-byte chunk (ds_read_b128 in
                                      // lane groups of 16) fall on 16 distinct 16-byte bank slots -- no swizzle, one address register
constexpr int X32_UNIT = X32_RT * X32_ROWB;   // bytes per ring slot: 32 whole row images
constexpr int X32_NW = 8;             // waves
constexpr int X32_KS = 32;            // k-steps of 32 bytes
constexpr int X32_AUXB = 8;           // per-unit aux buffers (256 B each; a unit's values live from its request to its screening)
constexpr int X32_QT = 256;           // queries per workgroup
constexpr int X32_WQ_CAP = 64;         // records per wave-private candidate queue (one ballot's worth always fits)
constexpr int x32_lds_bytes(int ns) { return ns * X32_UNIT + X32_AUXB * 256 + X32_NW * X32_WQ_CAP * 16 + 128; }   // (+ the FREE ring's two counters, + k_i8_filter_x32l's per-unit aux extremes)

// VAR bits (the shipped build instantiates one variant; the rest is the tuning build's, profiles/r03_c3_x32.txt, r05_c3_*.txt):
// 1 = EARLY (rows land one barrier early, fragments prefetched across the barrier), 2 = requests spread over the stream
// instead of behind the first MFMAs, 4 = SHIFT (the two waves of a SIMD half a unit apart: waves 4-7 meet the ring barrier in the
// MIDDLE of their stream, see below; needs X32_D <= X32_NS - 2), 8 = two accumulator chains (even / odd k-steps), 16 = s_setprio 1 for the
// second-dispatched half of the waves, 16384 = inline-asm fragment reads with counted waits, 32768 = only waves 0-3 request rows
// X32_NS ring slots, X32_D units requested ahead
template <int LK, int EPI, int VAR, int X32_NS = 4, int X32_D = 3>
__global__ __launch_bounds__(X32_NW * 64, 2) void k_i8_filter_x32(LowpParams P) {
    static_assert(X32_D >= 1 && X32_D < X32_NS && (!((VAR & 1) != 0) || X32_D >= 2), "ring geometry");
    // SHIFT: every wave still meets ONE s_barrier per unit, but waves 4-7 -- the second wave of each SIMD -- meet it between their
    // k-steps 15 and 16 instead of at the end of their stream.  In lock step both waves of a SIMD leave the matrix pipe idle through
    // the same tail (last reads returning, aux reads, frontier arithmetic, vmcnt wait), barrier and restart; half a unit apart, one
    // wave's tail and restart fall into the middle of the other's MFMA stream.  Ring: at barrier j the late waves are still reading
    // unit j's slot (and will until half a period later), so the requests of the early waves' unit j + 1 may only overwrite the slot
    // of unit j - 1: X32_D <= X32_NS - 2.
    constexpr bool SHIFT = (VAR & 4) != 0;
    static_assert(!SHIFT || X32_D <= X32_NS - 2, "SHIFT: the late waves read a unit's slot half a period past its barrier");
    static_assert(!SHIFT || (VAR & 32768), "SHIFT: the late waves must not request rows (they would have to wait for them at their barrier)");
    static_assert(LK == LP_I8 || LK == LP_U8, "int8 / uint8 rows with 4-byte aux values");
    // FREE (VAR bit 1024; TUNING BUILD ONLY, kept as the record of an experiment: no gain, and WRONG replies at small sizes -- an LDS-DMA
    // write is not ordered behind the requesting wave's ds_add the way it is behind a barrier, profiles/r05_c3_batches.txt): no s_barrier
    // in the loop at all.  The stamps of the lock-step kernel (profiles/r05_c3_batch1.txt) show
    // every wave 720-1140 of its ~4000 cycles per unit inside the ring barrier: eight waves whose unit times vary (request issue
    // blocked by the memory pipeline, arbitration for the matrix pipe, a candidate to queue) wait for the slowest of the eight,
    // every unit.  Two monotonic LDS counters carry what the barrier carried:
    //   landed += 1 by each of the 4 requesting waves once ITS rows of a unit are in LDS (its vmcnt says so): unit w may be read once
    //              landed >= 4 (w + 1);
    //   done   += 1 by each of the 8 waves once its last fragment read of a unit is in its LDS queue (the queue is served in order,
    //              so the add runs behind the reads): the slot of unit w may be overwritten once done >= 8 (w + 1).
    // A wave reads a counter where it drains lgkmcnt anyway (in front of the fragment reads of the next unit), and with the ring's
    // slack -- X32_D units requested ahead, X32_NS - X32_D - 1 slots for laggards -- the first look almost always passes, so nobody
    // spins; waves drift up to that slack apart, which also takes the two waves of a SIMD out of phase by itself.
    // No deadlock: the wave that is furthest behind (unit m) waits only for `landed`, i.e. for requesting waves to reach their
    // publication point in iteration m, and those wait only for done(m - 2), which every wave at or past unit m has signalled.
    constexpr bool FREE = (VAR & 1024) != 0;
#ifndef VSGPU_TUNING
    static_assert(!FREE, "FREE is a tuning-build experiment");
#endif
    static_assert(!FREE || ((VAR & 1) && (VAR & 32768) && !(VAR & 4) && !(VAR & 2)), "FREE: EARLY + requests by waves 0-3, no SHIFT / SPREAD");
    static_assert(!FREE || X32_D <= X32_NS - 2, "FREE: one slot of slack for laggards");
    constexpr bool EARLY = (VAR & 1) != 0;
    constexpr bool SPREAD = (VAR & 2) != 0;
    constexpr bool ACC2 = (VAR & 8) != 0;
    // diagnosis (replies meaningless): 32 = no row requests, 64 = no fragment reads / MFMAs, 128 = no screening, 256 = clocks:
    // wave 0 of every workgroup stores {s_memtime, s_memrealtime} deltas over the kernel into P.tilemin
    constexpr bool STAMPS = (VAR & 512) != 0;   // s_memtime sums per wave: {issue, rest of stream, vmcnt wait, barrier, units}
    constexpr bool NO_DMA = (VAR & 32) != 0, NO_MMA = (VAR & 64) != 0, NO_SCREEN = (VAR & 128) != 0, CLOCKS = (VAR & 256) != 0;
    // ASMRD: fragment reads in inline asm with counted lgkmcnt waits, 8 in flight (hipcc pairs its own reads with lgkmcnt(0):
    // one LDS round trip per two MFMAs, which is what the stream's time turned out to be made of)
    constexpr bool ASMRD = (VAR & 16384) != 0;
    constexpr int PF = ASMRD ? 8 : 4;   // A fragments in flight
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, h = lane >> 5;
    const int qtile = (int)blockIdx.y;
    const bool late_wave = SHIFT && wave >= X32_NW / 2;   // (wave-uniform: a scalar branch)

    i32x4_t qf[X32_KS];
    {
        const i32x4_t *src = reinterpret_cast<const i32x4_t *>(P.qfrag) + ((size_t)(qtile * X32_NW + wave) * X32_KS) * 64 + lane;
#pragma unroll
        for (int s = 0; s < X32_KS; s++) qf[s] = src[(size_t)s * 64];
    }
    const int qidx = qtile * X32_QT + wave * 32 + n32;
    uint32_t qaux = P.qaux[qidx];
    float tau = P.tau[qidx];
    // the ordinary loads are pinned before the first DMA (counted vmcnt below); the query fragments live in AGPRs, which the
    // MFMA reads directly
#pragma unroll
    for (int s = 0; s < X32_KS; s++) asm volatile("" : "+v"(qf[s]));
    asm volatile("" : "+v"(qaux), "+v"(tau));
    // int8 Cosine screen (see k_mfma_filter_lowp): dot >= ((1 - tau) - m) * nq * nx is necessary for score <= tau
    const float omt = 1.0f - tau;
    // (padding queries carry tau = -inf: nothing scores at or below it, +inf screens every row out)
    const float cosq = tau == -INFINITY ? INFINITY : (omt - 1e-5f * (1.0f + fabsf(omt))) * __uint_as_float(qaux);

    // staging: piece i of this wave is row 4 wave + i of the unit: one LDS-DMA instruction moves the row's 1 KiB (lane l its
    // 16-byte chunk l) to the row's image, rows X32_ROWB apart
    // ISS4: only waves 0 .. 3 -- the first-dispatched ones, which win the arbitration for the memory pipeline: their requests
    // cost a fraction of the younger waves' (s_memtime stamps) -- request rows, 8 pieces each
    constexpr int NISS = (VAR & 32768) ? 4 : X32_NW, IPWX = 32 / NISS;
    const bool issuer = NISS == X32_NW || wave < NISS;
    const uint32_t lds_wave_off = (uint32_t)(wave * IPWX * X32_ROWB);
    char *aux_lds = lds + X32_NS * X32_UNIT;
    // Candidates wait in a wave-private LDS queue as {row, query, dot, aux} -- the fill count is a scalar register, the slot of a
    // lane comes from the ballot (no LDS atomic, no return trip) -- and are finalised (the reference's epilogue with its IEEE
    // divide, the exact test against tau) when the queue is flushed: the stream's waves never spend more than a store on one.
    // The shared queue of the 16 x 16 x 64 kernel cost this kernel a third of its time: with ~5 K candidates per query some
    // wave of the eight met one in most units, and the barrier made the other seven wait for its atomics and divides.
    const uint32_t wq_off = mf_lds_offset(lds + X32_NS * X32_UNIT + X32_AUXB * 256) + (uint32_t)wave * (X32_WQ_CAP * 16);
    uint32_t wq_n = 0;
    const uint32_t aux_lds_off = mf_lds_offset(aux_lds);
    // FREE: the two counters behind the wave queues
    const uint32_t landed_off = mf_lds_offset(lds + X32_NS * X32_UNIT + X32_AUXB * 256 + X32_NW * X32_WQ_CAP * 16), done_off = landed_off + 32;
    auto ring_signal = [&](uint32_t off) {
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(off), "v"(1u) : "memory");
    };
    auto ring_wait = [&](uint32_t off, uint32_t target) {   // returns once the counter has reached `target`
        for (;;) {
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
            if ((int)((uint32_t)__builtin_amdgcn_readfirstlane((int)v) - target) >= 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    if (FREE) {
        if (tid == 0) {
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %2, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(landed_off), "v"(0u), "v"(done_off) : "memory");
        }
        __syncthreads();
    }
    if (VAR & 16) {
        if (wave >= X32_NW / 2) __builtin_amdgcn_s_setprio(1);
    }

    uint64_t clk0 = 0, rt0 = 0;
    uint64_t ph[6] = {0, 0, 0, 0, 0, 0}, t_prev = 0;
    auto stamp = [&](int i) {
        if (STAMPS) {
            const uint64_t now = __builtin_amdgcn_s_memtime();
            if (i >= 0) ph[i] += now - t_prev;
            t_prev = now;
        }
    };
    if (CLOCKS) {
        clk0 = __builtin_amdgcn_s_memtime();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    const uint32_t step = gridDim.x;
    auto tile_row0 = [&](uint32_t t) -> uint32_t {
        return (P.tile_first + (t >> P.tile_run_shift) * (P.tile_step << P.tile_run_shift) + (t & ((1u << P.tile_run_shift) - 1u))) * X32_RT;
    };
    // frontier = the unit requested next.  Everything about it is wave-uniform (row numbers, slab bases), so the requests
    // use the scalar-base form of global_load_lds: no address VGPRs beyond the lane's 16-byte chunk offset
    uint32_t f_r0 = 0;
    uint32_t cur_slab = 0xFFFFFFFFu;
    uint64_t cur_sbase = 0, cur_abase = 0;
    auto make_ptrs = [&](uint32_t t) {
        const uint32_t tt = t < P.n_tiles ? t : P.n_tiles - 1;   // requests past the last tile re-read it (never consumed)
        f_r0 = __builtin_amdgcn_readfirstlane(tile_row0(tt));
        const uint32_t sidx = f_r0 >> P.slab_shift;
        if (sidx != cur_slab) {
            cur_slab = sidx;
            const char *const *sp = P.slabs + sidx;
            const uint32_t *const *axp = P.aux_slabs + sidx;
            asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(cur_sbase), "=&s"(cur_abase)
                         : "s"(sp), "s"(axp)
                         : "memory");
        }
    };
    uint32_t ftile = blockIdx.x, fslot = 0, fbuf = 0;
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto issue_piece = [&](int i) {
        if (NO_DMA || !issuer) return;
        uint32_t row = f_r0 + (uint32_t)(IPWX * wave + i);
        if (row >= P.n_rows) row = P.n_rows - 1;
        const char *rowp = reinterpret_cast<const char *>(cur_sbase) + (size_t)(row & P.slab_mask) * P.row_stride;
        glds16<2>(rowp + lane16, fslot * X32_UNIT + lds_wave_off + (uint32_t)i * X32_ROWB, lds);
    };
    auto issue_aux = [&]() {
        if (wave == 0) {
            uint32_t arow = f_r0 + (uint32_t)lane;
            if (arow >= P.n_rows) arow = P.n_rows - 1;
            glds4(reinterpret_cast<const uint32_t *>(cur_abase) + (size_t)(arow & P.slab_mask), fbuf * 256u, aux_lds);
        }
    };
    auto advance_frontier = [&]() {
        ftile += step;
        fslot = fslot + 1 == X32_NS ? 0 : fslot + 1;
        fbuf = (fbuf + 1) & (X32_AUXB - 1);
        make_ptrs(ftile);
    };
    make_ptrs(ftile);
#pragma unroll
    for (int u = 0; u < X32_D; u++) {
        issue_aux();
#pragma unroll
        for (int i = 0; i < IPWX; i++) issue_piece(i);
        advance_frontier();
    }
    // per unit a wave has IPW row pieces in flight, wave 0 one aux piece more (requested first)
    auto wait_units_in_flight = [&](int units) {   // returns once at most `units` whole units of this wave's requests are outstanding
        if (wave == 0) lowp_wait_vmcnt(units * (IPWX + 1));
        else if (issuer) lowp_wait_vmcnt(units * IPWX);
    };
    if (FREE) {
        if (issuer) {   // unit 0 of this wave's rows has landed: publish it; iteration `it` publishes unit it + 1
            wait_units_in_flight(X32_D - 1);
            ring_signal(landed_off);
        }
        ring_wait(landed_off, 4u);
    } else {
        wait_units_in_flight(EARLY ? X32_D - 2 : X32_D - 1);
        mf_ring_barrier();
    }

    const uint32_t frag_lane_off = (uint32_t)n32 * X32_ROWB + (uint32_t)h * 16u;
    const uint32_t frag_lds_off = mf_lds_offset(lds) + frag_lane_off;
    auto read_frag = [&](uint32_t slot, int ks) -> i32x4_t {
        // row n32, bytes 32 ks + 16 h .. +16 of the row image
        i32x4_t v;
        if (ASMRD) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(frag_lds_off + slot * X32_UNIT), "n"(ks * 32));
        } else {
            v = *reinterpret_cast<const i32x4_t *>(lds + slot * X32_UNIT + frag_lane_off + ks * 32);
            if (LK == LP_U8) v ^= (int)0x80808080;
        }
        return v;
    };

    // the stream's test of one (row, query) value: passes every value the exact test passes (Cosine: one multiply against
    // cosq instead of the divide; NaN thresholds -- zero norms -- pass on to the exact test)
    auto screen_pass = [&](int dot, uint32_t av) -> bool {
        // (round 4 timing experiment: ONE integer compare per value here instead of cvt + mul + cmp moved config 3's kernel from
        // 13.43 to 13.25 ms -- a per-unit norm extreme to make that exact is not worth its bookkeeping)
        if (EPI == LE_I8_COS) return !((float)dot < cosq * __uint_as_float(av));
        if (EPI == LE_I8_L2) return (float)((int)av + (int)qaux - 2 * dot) <= tau;
        if (EPI == LE_I8_IP) return (float)(1 - dot) <= tau;
        return (float)(1 - (dot + 128 * (int)av + (int)qaux)) <= tau;
    };
    // queue -> the per-query candidate lists: exact score of every record, in the reference's operation order
    auto flush_wave_queue = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if ((uint32_t)lane < wq_n) {
            u32x4_t rec;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rec) : "v"(wq_off + (uint32_t)lane * 16u) : "memory");
            const uint32_t row = rec[0], q = rec[1], av = rec[3];
            const int dot = (int)rec[2];
            const uint32_t qa = P.qaux[q];
            const float tq = P.tau[q];
            float sc;
            if (EPI == LE_I8_L2) sc = (float)((int)av + (int)qa - 2 * dot);
            else if (EPI == LE_I8_IP) sc = (float)(1 - dot);
            else if (EPI == LE_U8_IP) sc = (float)(1 - (dot + 128 * (int)av + (int)qa));
            else sc = __fsub_rn(1.0f, __fdiv_rn((float)dot, __fmul_rn(__uint_as_float(av), __uint_as_float(qa))));
            if (sc <= tq) {
                const uint32_t s = atomicAdd(&P.counts[q], 1u);
                if (s < P.cap) P.cand[(size_t)q * P.cap + s] = make_uint2(row, __float_as_uint(sc));
            }
        }
        // (the loads, atomics and stores above share the VM counter with the row requests: drain, the counted waits stay valid)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        wq_n = 0;
    };

    i32x16_t acc_prev;
#pragma unroll
    for (int i = 0; i < 16; i++) acc_prev[i] = 0;
    // aux values of the unit being screened: 4 x 4 consecutive rows, rows 8 g + 4 h + (0..3); read (inline asm: a
    // compiler-visible read of a DMA target draws a vmcnt(0)) right before the barrier that ends the unit they belong to
    u32x4_t auxv[4];
#pragma unroll
    for (int g = 0; g < 4; g++) auxv[g] = u32x4_t{0, 0, 0, 0};
    uint32_t r0_prev = 0, nvalid_prev = 0;
    uint32_t cslot = 0, cbuf = 0, units_done = 0;
    i32x4_t afr[PF];
    if (EARLY) {
#pragma unroll
        for (int f = 0; f < PF; f++) afr[f] = read_frag(0, f);
        if (ASMRD) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int f = 0; f < PF; f++) asm volatile("" : "+v"(afr[f]));
        }
    }

    // one iteration past the last tile: its MFMAs run on the clamped refill (discarded), its screening is the last tile's
    uint32_t tile = blockIdx.x;
    const uint32_t my_tiles = tile < P.n_tiles ? (P.n_tiles - tile + step - 1) / step : 0;
    for (uint32_t it = 0; it <= my_tiles; it++, tile += step) {
        const uint32_t nslot = cslot + 1 == X32_NS ? 0 : cslot + 1;
        i32x16_t acc, acc_b;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0, acc_b[i] = 0;
        stamp(it ? 4 : -1);
        if (FREE) {
            // the requests of this iteration (unit it + D) overwrite the slot of unit it + D - NS: every wave is done with it
            if (issuer && it + X32_D >= (uint32_t)X32_NS) ring_wait(done_off, 8u * (it + X32_D - X32_NS + 1));
        }
        issue_aux();   // (a wave-uniform branch: kept out of the stream's scheduling region)
        if (!EARLY) {
#pragma unroll
            for (int f = 0; f < PF; f++) afr[f] = read_frag(cslot, f);
        }
        unsigned long long any = 0, any_g[4] = {0, 0, 0, 0};   // lanes holding a survivor of unit u-1, per group of four accumulator registers (v_cmp + s_or per value)
        // ---- the unit's stream: 32 x {MFMA, next fragment read}, the refill requests for unit u + D (into the slot of unit
        // u - 1, free since the barrier) and the screening of unit u - 1 in between ----
        {
            constexpr int PBASE = SPREAD ? 2 : 1, PSTEP = SPREAD ? 32 / IPWX : 1;
#pragma unroll
            for (int ks = 0; ks < X32_KS; ks++) {
                if (SHIFT && ks == X32_KS / 2) {
                    // the late waves' barrier of this unit (they request no rows: no vmcnt to wait for)
                    if (late_wave) {
                        stamp(5);
                        mf_ring_barrier();
                        stamp(3);
                    }
                }
                if (!NO_MMA) {
                    if (ASMRD) {
                        // fragment ks has returned once at most this many later reads are outstanding (LDS returns in order;
                        // EARLY: fragments 0 .. PF-1 came back before the barrier)
                        constexpr int BASE = 0;
                        const int left = EARLY ? PF - 1 : (X32_KS - 1 - ks < PF - 1 ? X32_KS - 1 - ks : PF - 1);
                        if (!(EARLY && ks < PF)) lowp_wait_lgkmcnt(left + BASE, afr[ks % PF]);
                        if (LK == LP_U8) afr[ks % PF] ^= (int)0x80808080;
                    }
                    const i32x4_t a = afr[ks % PF];
                    if (ACC2 && (ks & 1)) acc_b = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[ks], acc_b, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[ks], acc, 0, 0, 0);
                    const int f = ks + PF;
                    if (FREE && f == X32_KS) {
                        // every fragment read of this unit is in the LDS queue: the slot is done with, as far as this wave goes;
                        // the next unit's rows must have landed before its first fragments are read
                        ring_signal(done_off);
                        ring_wait(landed_off, 4u * (it + 2));
                    }
                    if (f < X32_KS) afr[ks % PF] = read_frag(cslot, f);
                    else if (EARLY) afr[ks % PF] = read_frag(nslot, f - X32_KS);
                }
                if (ks >= PBASE && (ks - PBASE) % PSTEP == 0 && (ks - PBASE) / PSTEP < IPWX) issue_piece((ks - PBASE) / PSTEP);
                if (FREE && ks == PBASE + IPWX + 3) {
                    // a few MFMAs behind this iteration's requests: the rows of unit it + 1 (requested one iteration ago) have
                    // landed once only this iteration's requests are outstanding -- publish them
                    if (issuer) {
                        wait_units_in_flight(X32_D - 1);
                        ring_signal(landed_off);
                    }
                }
                if (STAMPS && !SPREAD && ks == PBASE + IPWX - 1) stamp(0);
                // screening of unit u-1, one accumulator register per k-step
                if (!NO_SCREEN && ks >= 8 && ks < 24) {
                    const int r = ks - 8;
                    const uint32_t av = auxv[r >> 2][r & 3];
                    const int dot = acc_prev[r];
                    any_g[r >> 2] |= __ballot(screen_pass(dot, av));
                }
            }
            // the interleave, pinned (left alone hipcc hoists the screening to the front and reads fragments two at a time
            // right before their MFMAs): per k-step one MFMA, the read of fragment ks + PF, the request that belongs
            // there, and the step's share of the VALU work
#pragma unroll
            for (int ks = 0; ks < X32_KS; ks++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (!ASMRD && (ks + PF < X32_KS || EARLY)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (ks >= PBASE && (ks - PBASE) % PSTEP == 0 && (ks - PBASE) / PSTEP < IPWX) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, LK == LP_U8 ? 8 : 4, 0);
            }
        }
        stamp(1);
        // some lane of the wave holds a candidate of unit u-1: queue it (per group of four registers -- usually one group, one value)
        any = any_g[0] | any_g[1] | any_g[2] | any_g[3];
        if (any != 0 && !(NO_DMA || NO_MMA || NO_SCREEN)) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (any_g[g] == 0) continue;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int r = 4 * g + rr;
                    const uint32_t lrow = (uint32_t)(rr + 8 * g + 4 * h);
                    const uint32_t av = auxv[g][rr];
                    const int dot = acc_prev[r];
                    const bool hit = screen_pass(dot, av) && lrow < nvalid_prev;
                    const unsigned long long b = __ballot(hit);
                    if (b == 0) continue;
                    const uint32_t nb = (uint32_t)__builtin_popcountll(b);
                    if (wq_n + nb > (uint32_t)X32_WQ_CAP) flush_wave_queue();
                    if (hit) {
                        const uint32_t pos = wq_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                        const u32x4_t rec = {r0_prev + lrow, (uint32_t)qidx, (uint32_t)dot, av};
                        asm volatile("ds_write_b128 %0, %1" ::"v"(wq_off + pos * 16u), "v"(rec) : "memory");
                    }
                    wq_n += nb;
                }
            }
        }
        // this unit becomes the one screened next
        if (ACC2) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc_prev[i] = acc[i] + acc_b[i];
        } else acc_prev = acc;
        if (NO_DMA || NO_MMA || NO_SCREEN) asm volatile("" ::"v"(acc_prev), "s"(any));   // (keeps the diagnosis variants' work alive)
        r0_prev = tile_row0(tile < P.n_tiles ? tile : P.n_tiles - 1);
        nvalid_prev = it < my_tiles ? P.n_rows - r0_prev : 0;
        {
            const uint32_t aoff = aux_lds_off + cbuf * 256u + (uint32_t)h * 16u;
#pragma unroll
            for (int g = 0; g < 4; g++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(auxv[g]) : "v"(aoff), "n"(g * 32));
        }
        cbuf = (cbuf + 1) & (X32_AUXB - 1);
        cslot = nslot;
        units_done++;
        advance_frontier();
        // the next unit (EARLY: the one after it as well) has landed; every read of this unit's slot, and the aux reads
        // above, have returned (mf_ring_barrier waits lgkmcnt(0))
        if (!late_wave && !FREE) {
            stamp(5);
            wait_units_in_flight(EARLY ? X32_D - 2 : X32_D - 1);
            stamp(2);
            mf_ring_barrier();
            stamp(3);
        }
        asm volatile("" : "+v"(auxv[0]), "+v"(auxv[1]), "+v"(auxv[2]), "+v"(auxv[3]));
        if (ASMRD && EARLY) {
#pragma unroll
            for (int f = 0; f < PF; f++) asm volatile("" : "+v"(afr[f]));
        }
    }
    if (STAMPS && lane == 0) {
        uint64_t *o = reinterpret_cast<uint64_t *>(P.tilemin) + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * X32_NW + wave) * 8;
        for (int i = 0; i < 4; i++) o[i] = ph[i];
        o[4] = units_done;
        o[5] = ph[4];
        o[6] = ph[5];
    }
    if (CLOCKS && tid == 0) {
        const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        uint64_t *o = reinterpret_cast<uint64_t *>(P.tilemin) + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2;
        o[0] = c1 - clk0;
        o[1] = r1 - rt0;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (wq_n) flush_wave_queue();
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_i8_filter_x32l -- the same filter with a THIRD fewer instructions per unit (round 5).
//
// What round 5 measured on k_i8_filter_x32 (profiles/r05_c3_batch1.txt, r05_c3_batch2.txt): the kernel is bound by instruction
// ISSUE, not by a pipe.  Two waves share a SIMD's issue port; per 32-row unit they issue ~760 instructions for 64 MFMAs (2048
// cycles of matrix pipe), and every elimination run moves the time in proportion to the instructions it removes (screening off:
// -17 % instructions, -22 % time; no row requests: -8 %, -27 % with their issue stalls) while removing the ring barrier altogether
// (two LDS counters instead, VAR bit 1024) or prefetching deeper changes nothing.  So this kernel issues less:
//   * screening: ONE integer threshold per query for the whole scan -- from the table-wide extremes of the rows' aux values
//     (LowpParams::sq8_max, k_row_aux_i8) -- and a running maximum over the 16 accumulators of the previous unit (8 v_max3_i32
//     + 1 compare per unit, was 48 VALU + 20 SALU).  A lane whose maximum reaches the threshold sends the wave into the exact
//     per-value test (the reference's epilogue order, as before), which reads the unit's aux values only then;
//   * the accumulators alternate between two register sets from unit to unit instead of being copied (16 v_mov per unit);
//   * row requests take the row's address from a scalar base that is bumped by the row stride (global_load_lds with an SGPR
//     base: 5 scalar instructions per 1 KiB piece, was 8 of which 2 VALU), and the requesting waves (0-3) run their own copy of
//     the loop: the other four never see a request or the branch around it;
//   * tile -> row arithmetic is the filter's (consecutive tiles), no probe strides.
// Thresholds (T = smallest dot that can still pass; every value the exact test passes has dot >= T):
//   Cosine   exact test  !(float(dot) < cosq * norm_x)            T = ceil(cosq * (cosq >= 0 ? min norm : max norm))
//   L2       float(sum_x + sum_q - 2 dot) <= tau                  T = ceil((min sum_x + sum_q - tau') / 2)
//   IP       float(1 - dot) <= tau                                T = ceil(1 - tau')
//   uint8 IP float(1 - (dot + 128 s_x + c_q)) <= tau              T = ceil(1 - tau' - 128 max s_x - c_q)
// with tau' = tau + |tau| 2^-22 (the int -> float conversion of the left side rounds to nearest: monotone, relative error 2^-24).
// A table whose aux values spread widely gets a loose threshold and takes the exact test often: vsgpu_lowp.hip picks
// k_i8_filter_x32 (per-value screen in the stream) for those.
// VAR bits: 4 = SHIFT (as above), 1 = EARLY is always on; ring X32_NS / X32_D as above.
template <int EPI> __device__ static inline int x32l_threshold(float tau, uint32_t qaux, int ext_min, int ext_max) {
    if (tau != tau) return (int)0x80000000;                       // NaN threshold: everything goes to the exact test
    if (EPI == LE_I8_COS) {
        if (tau == -INFINITY) return 0x7FFFFFFF;                     // padding query
        const float omt = 1.0f - tau;
        const float cosq = (omt - 1e-5f * (1.0f + fabsf(omt))) * __uint_as_float(qaux);   // (k_i8_filter_x32's cosq)
        if (cosq != cosq) return (int)0x80000000;
        const float t = cosq * __int_as_float(cosq >= 0.0f ? ext_min : ext_max);
        if (t != t) return (int)0x80000000;
        if (t >= 2147483520.0f) return 0x7FFFFFFF;
        if (t <= -2147483520.0f) return (int)0x80000000;
        return (int)ceilf(t);
    }
    const double tb = (double)tau + fabs((double)tau) * (1.0 / 4194304.0);
    double t;
    if (EPI == LE_I8_L2) t = ((double)ext_min + (double)(int)qaux - tb) * 0.5;
    else if (EPI == LE_I8_IP) t = 1.0 - tb;
    else t = 1.0 - tb - 128.0 * (double)ext_max - (double)(int)qaux;
    if (t >= 2147483000.0) return 0x7FFFFFFF;
    if (t <= -2147483000.0) return (int)0x80000000;
    return (int)ceil(t);
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int LK, int EPI, int VAR, int X32_NS = 4, int X32_D = 3>
__global__ __launch_bounds__(X32_NW * 64, 2) void k_i8_filter_x32l(LowpParams P) {
    static_assert(X32_D >= 2 && X32_D < X32_NS, "ring geometry (rows land one barrier early)");
    static_assert(LK == LP_I8 || LK == LP_U8, "int8 / uint8 rows with 4-byte aux values");
    constexpr bool SHIFT = (VAR & 4) != 0;
    static_assert(!SHIFT || X32_D <= X32_NS - 2, "SHIFT: the late waves read a unit's slot half a period past its barrier");
    // ASMRD (VAR bit 16384): fragment reads in inline asm with counted lgkmcnt waits (hipcc keeps two reads ahead of
    // their MFMAs and waits out an LDS round trip per pair: with two waves per SIMD that alone caps the matrix pipe near 60 %)
    constexpr bool ASMRD = (VAR & 16384) != 0;
    // NODEF (VAR bit 8): no deferred screening -- the unit's own accumulators are reduced and tested behind its last MFMA, so one
    // accumulator set (16 registers) goes and 8 fragments in flight fit without spilling
    constexpr bool NODEF = (VAR & 8) != 0;
    // diagnosis (replies meaningless; tuning build): 128 = the threshold test never fires, 64 = no aux requests
    // QDEFER (VAR bit 32): a value at or above the integer threshold is QUEUED as it is -- {row, query, dot} --, and the queue's flush
    // fetches the row's aux value from global memory for the exact score.  The stream then needs no aux values at all (no aux
    // request per unit, no LDS round trip in front of the per-value test: with ~0.3 values per wave and unit over the threshold
    // some wave of the eight took that round trip in nearly every unit, and the ring barrier made the other seven wait for it).
    constexpr bool QDEFER = (VAR & 32) != 0;
    constexpr bool NO_TEST = (VAR & 128) != 0 && (VAR & 512) == 0, NO_AUX = (VAR & 64) != 0 || QDEFER;
    constexpr bool NEVER = (VAR & 128) != 0 && (VAR & 512) != 0;
    constexpr bool COUNT = (VAR & 1024) != 0;   // tuning build: every wave prints how many of its units sent it into the exact test
    uint32_t n_fired = 0, n_groups = 0, n_units = 0;   // the test and the exact path stay in the code, the threshold is unreachable
    // UNITTHR (VAR bit 256; Cosine, with ASMRD; tuning build only -- measured: the exact test fires on 57 % instead of 73 % of the
    // wave-units at 8 Mi rows, and the kernel time does not move, 12.3 against 12.2 ms at 50 M rows: what the tighter threshold saves,
    // its own bookkeeping costs, profiles/r05_c3_batches.txt): the threshold of a unit comes from the extremes of ITS 32 norms instead of the
    // table's (over 50 M uniform rows the table-wide minimum sits 7.7 % under the typical norm, a unit's 2.8 %: the exact test fires
    // on 14 % of the wave-units instead of 27 %).  Wave 7 -- it requests no rows and meets the barrier mid-stream: the wave with the
    // most slack -- reduces the next unit's aux values (DPP butterflies) and leaves {min, max} in an 8-deep LDS ring; every wave
    // reads the pair of the unit under test early in its stream and rebuilds its threshold with five VALU operations.
    constexpr bool UNITTHR = (VAR & 256) != 0 && EPI == LE_I8_COS && ASMRD && !NO_AUX;
    constexpr int PF = (ASMRD && NODEF) ? 8 : 4;   // A fragments in flight; must divide the 32 k-steps (fragment f lives in afr[f % PF] in
                                                   // every unit); 8 with two accumulator sets would spill (scratch traffic would also
                                                   // break the counted vmcnt waits)
    static_assert(X32_KS % PF == 0, "fragment ring");
    constexpr int IPWX = 8;        // row pieces per requesting wave (waves 0-3) and unit
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n32 = lane & 31, h = lane >> 5;
    const int qtile = (int)blockIdx.y;
    const bool late_wave = SHIFT && wave >= X32_NW / 2;

    i32x4_t qf[X32_KS];
    {
        const i32x4_t *src = reinterpret_cast<const i32x4_t *>(P.qfrag) + ((size_t)(qtile * X32_NW + wave) * X32_KS) * 64 + lane;
#pragma unroll
        for (int s = 0; s < X32_KS; s++) qf[s] = src[(size_t)s * 64];
    }
    const int qidx = qtile * X32_QT + wave * 32 + n32;
    uint32_t qaux = P.qaux[qidx];
    float tau = P.tau[qidx];
    const int ext_min = (int)P.sq8_max[0], ext_max = (int)P.sq8_max[1];
#pragma unroll
    for (int s = 0; s < X32_KS; s++) asm volatile("" : "+v"(qf[s]));
    asm volatile("" : "+v"(qaux), "+v"(tau));
    int thr = x32l_threshold<EPI>(tau, qaux, ext_min, ext_max);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(thr));   // (every ordinary load has returned: the counted vmcnt waits below see DMA only)
    const float omt = 1.0f - tau;
    const float cosq = tau == -INFINITY ? INFINITY : (omt - 1e-5f * (1.0f + fabsf(omt))) * __uint_as_float(qaux);
    if (NEVER) {
        thr = 0x7FFFFFFF;
        asm volatile("" : "+v"(thr));
    }
    // UNITTHR: lanes whose threshold is a constant whatever the rows (padding queries, NaN thresholds) keep it
    const int thr_table = thr;
    const bool thr_fixed = thr == (int)0x80000000 || thr == 0x7FFFFFFF;
    const bool cosq_neg = cosq < 0.0f;

    const uint32_t lds_base = mf_lds_offset(lds);
    const uint32_t aux_lds_off = lds_base + X32_NS * X32_UNIT;
    const uint32_t wq_off = aux_lds_off + X32_AUXB * 256 + (uint32_t)wave * (X32_WQ_CAP * 16);
    uint32_t wq_n = 0;
    const uint32_t lane16 = (uint32_t)lane * 16u, lane4 = (uint32_t)lane * 4u;
    const uint32_t ext_off = aux_lds_off + X32_AUXB * 256 + X32_NW * X32_WQ_CAP * 16 + 64;   // UNITTHR: [8] x {min bits, max bits}
    // wave 7: {min, max} of the norms of the valid rows of the unit whose aux values sit in ring buffer `buf` -> ext ring
    auto reduce_unit_aux = [&](uint32_t buf, uint32_t nvalid) {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(aux_lds_off + buf * 256u + (uint32_t)(lane & 31) * 4u) : "memory");
        const bool ok = (uint32_t)(lane & 31) < nvalid;
        int lo = ok ? (int)v : 0x7F800000, hi = ok ? (int)v : 0;   // (norms are non-negative floats: their bits order like ints)
        // butterflies within rows of 16 lanes: quad xor 1, quad xor 2, half-row mirror, row mirror
        lo = min(lo, __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false));
        hi = max(hi, __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false));
        lo = min(lo, __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false));
        hi = max(hi, __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false));
        lo = min(lo, __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false));
        hi = max(hi, __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false));
        lo = min(lo, __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false));
        hi = max(hi, __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false));
        const int lo2 = min(__builtin_amdgcn_readlane(lo, 0), __builtin_amdgcn_readlane(lo, 16));
        const int hi2 = max(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(hi, 16));
        if (lane == 0) {
            const u32x2l_t e = {(uint32_t)lo2, (uint32_t)hi2};
            asm volatile("ds_write_b64 %0, %1" ::"v"(ext_off + buf * 8u), "v"(e) : "memory");
        }
    };
    // one 1 KiB row piece / one 256-byte aux piece: global address = scalar base + the lane's offset, LDS address = M0 + the lane's
    auto dma16 = [&](uint64_t sbase, uint32_t lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(lane16), "s"(sbase), "s"(lds_addr) : "memory", "m0");
    };
    auto dma4 = [&](uint64_t sbase, uint32_t lds_addr) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(lane4), "s"(sbase), "s"(lds_addr) : "memory", "m0");
    };

    const uint32_t step = gridDim.x;
    // frontier = the unit requested next (wave-uniform state, all of it in scalar registers)
    uint32_t cur_slab = 0xFFFFFFFFu;
    uint64_t cur_sbase = 0, cur_abase = 0;
    uint32_t ftile = blockIdx.x, fslot = 0, fbuf = 0;
    auto request_unit = [&](auto piece_filter) {
        // rows of tile ftile (clamped to the last tile: requests past the end re-read it, never consumed) -> slot fslot
        const uint32_t tt = ftile < P.n_tiles ? ftile : P.n_tiles - 1;
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((P.tile_first + tt) * X32_RT));
        const uint32_t sidx = r0 >> P.slab_shift;
        if (sidx != cur_slab) {
            cur_slab = sidx;
            const char *const *sp = P.slabs + sidx;
            const uint32_t *const *axp = P.aux_slabs + sidx;
            asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(cur_sbase), "=&s"(cur_abase)
                         : "s"(sp), "s"(axp)
                         : "memory");
        }
        piece_filter(r0);
    };
    auto advance_frontier = [&]() {
        ftile += step;
        fslot = fslot + 1 == X32_NS ? 0 : fslot + 1;
        fbuf = (fbuf + 1) & (X32_AUXB - 1);
    };
    // per unit: wave 0 the aux piece, waves 0-3 eight row pieces each (piece i of wave w = row 8 w + i of the unit)
    // Rows past the table's end (the last, partial tile) and tiles past the last one are READ like any other -- a slab is allocated
    // whole (vsgpu.hip ensure_rows: slab_rows * row_bytes + 1 KiB), tiles never straddle slabs -- and ignored: nvalid masks them.
    uint64_t pbase = 0;        // global address of this wave's next row piece
    uint32_t plds = 0;         // its LDS address
    auto begin_requests = [&]() {
        request_unit([&](uint32_t r0) {
            if (wave == 0 && !NO_AUX) {
                // 64 aux values from row r0 on: the unit's 32 at lanes 0..31, the next unit's (unused) behind them -- only at the end
                // of the table or of a slab (the aux slabs carry no padding) do the lanes need the clamp
                if (r0 + 64 <= P.n_rows && (r0 & P.slab_mask) + 64 <= P.slab_mask + 1u) dma4(cur_abase + (uint64_t)(r0 & P.slab_mask) * 4u, aux_lds_off + fbuf * 256u);
                else {
                    uint32_t arow = r0 + (uint32_t)lane;
                    if (arow >= P.n_rows) arow = P.n_rows - 1;
                    glds4(reinterpret_cast<const uint32_t *>(cur_abase) + (size_t)(arow & P.slab_mask), fbuf * 256u, lds + X32_NS * X32_UNIT);
                }
            }
            pbase = cur_sbase + (uint64_t)((r0 & P.slab_mask) + (uint32_t)(IPWX * wave)) * P.row_stride;
            plds = lds_base + fslot * X32_UNIT + (uint32_t)(wave * IPWX * X32_ROWB);
        });
    };
    auto issue_piece = [&]() {
        dma16(pbase, plds);
        pbase += P.row_stride;
        plds += X32_ROWB;
    };
    const bool issuer = wave < 4;
    if (issuer) {
#pragma unroll
        for (int u = 0; u < X32_D; u++) {
            begin_requests();
#pragma unroll
            for (int i = 0; i < IPWX; i++) issue_piece();
            advance_frontier();
        }
    }
    auto wait_units_in_flight = [&](int units) {
        if (wave == 0 && !NO_AUX) lowp_wait_vmcnt(units * (IPWX + 1));
        else if (issuer) lowp_wait_vmcnt(units * IPWX);
    };
    wait_units_in_flight(X32_D - 2);
    mf_ring_barrier();
    if (UNITTHR && wave == X32_NW - 1) {
        const uint32_t t0 = blockIdx.x < P.n_tiles ? blockIdx.x : P.n_tiles - 1;
        reduce_unit_aux(0u, P.n_rows - (P.tile_first + t0) * X32_RT);
    }

    const uint32_t frag_lane_off = (uint32_t)n32 * X32_ROWB + (uint32_t)h * 16u;
    const uint32_t frag_lds_off = lds_base + frag_lane_off;
    auto read_frag = [&](uint32_t slot, int ks) -> i32x4_t {
        i32x4_t v;
        if (ASMRD) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(frag_lds_off + slot * X32_UNIT), "n"(ks * 32));
        } else {
            v = *reinterpret_cast<const i32x4_t *>(lds + slot * X32_UNIT + frag_lane_off + ks * 32);
            if (LK == LP_U8) v ^= (int)0x80808080;
        }
        return v;
    };
    auto screen_pass = [&](int dot, uint32_t av) -> bool {
        if (EPI == LE_I8_COS) return !((float)dot < cosq * __uint_as_float(av));
        if (EPI == LE_I8_L2) return (float)((int)av + (int)qaux - 2 * dot) <= tau;
        if (EPI == LE_I8_IP) return (float)(1 - dot) <= tau;
        return (float)(1 - (dot + 128 * (int)av + (int)qaux)) <= tau;
    };
    auto flush_wave_queue = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if ((uint32_t)lane < wq_n) {
            u32x4_t rec;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rec) : "v"(wq_off + (uint32_t)lane * 16u) : "memory");
            const uint32_t row = rec[0], q = rec[1];
            uint32_t av = rec[3];
            const int dot = (int)rec[2];
            if (QDEFER && EPI != LE_I8_IP) av = P.aux_slabs[row >> P.slab_shift][row & P.slab_mask];
            const uint32_t qa = P.qaux[q];
            const float tq = P.tau[q];
            float sc;
            if (EPI == LE_I8_L2) sc = (float)((int)av + (int)qa - 2 * dot);
            else if (EPI == LE_I8_IP) sc = (float)(1 - dot);
            else if (EPI == LE_U8_IP) sc = (float)(1 - (dot + 128 * (int)av + (int)qa));
            else sc = __fsub_rn(1.0f, __fdiv_rn((float)dot, __fmul_rn(__uint_as_float(av), __uint_as_float(qa))));
            if (sc <= tq) {
                const uint32_t s = atomicAdd(&P.counts[q], 1u);
                if (s < P.cap) P.cand[(size_t)q * P.cap + s] = make_uint2(row, __float_as_uint(sc));
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        wq_n = 0;
    };
    // the exact per-value test of ONE group of four accumulators of a unit (rare: some lane's maximum over the group reached its
    // threshold): rows 8 g + 4 h + (0..3) of the unit, whose aux values are one 16-byte read
    auto exact_group = [&](const i32x16_t &acc, int g, uint32_t r0, uint32_t nvalid, uint32_t buf) {
        u32x4_t auxv = {0u, 0u, 0u, 0u};
        if (!QDEFER)
            asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(auxv) : "v"(aux_lds_off + buf * 256u + (uint32_t)h * 16u), "n"(g * 32) : "memory");
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const uint32_t lrow = (uint32_t)(rr + 8 * g + 4 * h);
            const uint32_t av = auxv[rr];
            const int dot = acc[4 * g + rr];
            const bool hit = (QDEFER ? dot >= thr : screen_pass(dot, av)) && lrow < nvalid;
            const unsigned long long b = __ballot(hit);
            if (b == 0) continue;
            const uint32_t nb = (uint32_t)__builtin_popcountll(b);
            if (wq_n + nb > (uint32_t)X32_WQ_CAP) flush_wave_queue();
            if (hit) {
                const uint32_t pos = wq_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
                const u32x4_t rec = {r0 + lrow, (uint32_t)qidx, (uint32_t)dot, av};
                asm volatile("ds_write_b128 %0, %1" ::"v"(wq_off + pos * 16u), "v"(rec) : "memory");
            }
            wq_n += nb;
        }
    };

    i32x4_t afr[PF];
#pragma unroll
    for (int f = 0; f < PF; f++) afr[f] = read_frag(0, f);
    if (ASMRD) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int f = 0; f < PF; f++) asm volatile("" : "+v"(afr[f]));
    }

    uint32_t r0_prev = 0, nvalid_prev = 0, buf_prev = 0;
    uint32_t cslot = 0, cbuf = 0;
    uint32_t tile = blockIdx.x;
    const uint32_t my_tiles = tile < P.n_tiles ? (P.n_tiles - tile + step - 1) / step : 0;

    // one unit: `acc` takes this unit's products, `prev` (the unit before) is screened meanwhile.  ISS = this wave requests rows.
    auto unit = [&](auto iss_c, i32x16_t &acc, const i32x16_t &prev, bool last) __attribute__((always_inline)) {
        constexpr bool ISS = decltype(iss_c)::value;
        const uint32_t nslot = cslot + 1 == X32_NS ? 0 : cslot + 1;
        if (ISS) begin_requests();
        u32x2l_t uext = {0u, 0u};
        if (UNITTHR) {
            if (!ISS && wave == X32_NW - 1) {   // the next unit's rows (and aux values) landed a barrier ago
                const uint32_t tn = tile + step < P.n_tiles ? tile + step : P.n_tiles - 1;
                reduce_unit_aux((cbuf + 1) & (X32_AUXB - 1), P.n_rows - (P.tile_first + tn) * X32_RT);
            }
            // {min, max} of the unit under test: the read returns in order with the fragment reads, long before the stream ends
            asm volatile("ds_read_b64 %0, %1" : "=v"(uext) : "v"(ext_off + buf_prev * 8u));
        }
        int mg[4] = {(int)0x80000000, (int)0x80000000, (int)0x80000000, (int)0x80000000};
#pragma unroll
        for (int ks = 0; ks < X32_KS; ks++) {
            if (SHIFT && ks == X32_KS / 2) {
                if (late_wave) mf_ring_barrier();
            }
            if (ASMRD) {
                // fragment ks has returned once at most PF - 1 later reads are outstanding (the LDS queue is served in order; other
                // LDS operations in the queue only make the wait stricter)
                lowp_wait_lgkmcnt(PF - 1, afr[ks % PF]);
                if (LK == LP_U8) afr[ks % PF] ^= (int)0x80808080;
            }
            const i32x4_t a = afr[ks % PF];
            if (ks == 0) {
                i32x16_t z;
#pragma unroll
                for (int i = 0; i < 16; i++) z[i] = 0;
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[ks], z, 0, 0, 0);
            } else acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[ks], acc, 0, 0, 0);
            const int f = ks + PF;
            afr[ks % PF] = f < X32_KS ? read_frag(cslot, f) : read_frag(nslot, f - X32_KS);
            if (ISS && ks >= 1 && ks <= IPWX) issue_piece();
            if (!NODEF && ks >= 12 && ks < 20) {   // the previous unit's accumulators: maxima of the four groups of four, two instructions each
                const int g = (ks - 12) >> 1;
                if ((ks & 1) == 0) mg[g] = max(max(prev[4 * g], prev[4 * g + 1]), prev[4 * g + 2]);
                else mg[g] = max(mg[g], prev[4 * g + 3]);
            }
        }
        // the interleave, pinned (left alone hipcc reads two fragments right in front of their two MFMAs and waits out the LDS
        // round trip): per k-step one MFMA, the read of fragment ks + PF, at most one VALU
#pragma unroll
        for (int ks = 0; ks < X32_KS; ks++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (!ASMRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, LK == LP_U8 ? 5 : 1, 0);
        }
        int thr_u = thr;
        if (UNITTHR) {
            // (the counted fragment waits have long passed the pair's read: at most PF - 1 younger reads were outstanding by k-step PF)
            asm volatile("" : "+v"(uext));
            const float t = cosq * __uint_as_float(cosq_neg ? uext[1] : uext[0]);
            thr_u = thr_fixed ? thr_table : (int)ceilf(t);
        }
        const uint32_t tt = tile < P.n_tiles ? tile : P.n_tiles - 1;
        if (NODEF) {   // this unit's own accumulators, behind its last MFMA
            r0_prev = (P.tile_first + tt) * X32_RT;
            nvalid_prev = last ? 0u : P.n_rows - r0_prev;
            buf_prev = cbuf;
#pragma unroll
            for (int g = 0; g < 4; g++) mg[g] = max(max(acc[4 * g], acc[4 * g + 1]), max(acc[4 * g + 2], acc[4 * g + 3]));
        }
        // a lane of the wave may hold a candidate of the unit under test: the exact test, value by value, group by group
        const int m = max(max(mg[0], mg[1]), max(mg[2], mg[3]));
        if (NO_TEST) asm volatile("" ::"v"(m));
        if (COUNT) n_units++;
        if (!NO_TEST && __ballot(m >= thr_u) != 0) {
            if (COUNT) n_fired++;
#pragma unroll
            for (int g = 0; g < 4; g++)
                if (__ballot(mg[g] >= thr_u) != 0) {
                    if (COUNT) n_groups++;
                    exact_group(NODEF ? acc : prev, g, r0_prev, nvalid_prev, buf_prev);
                }
        }
        r0_prev = (P.tile_first + tt) * X32_RT;
        nvalid_prev = last ? 0u : P.n_rows - r0_prev;
        buf_prev = cbuf;
        cbuf = (cbuf + 1) & (X32_AUXB - 1);
        cslot = nslot;
        tile += step;
        if (ISS) advance_frontier();
        if (!late_wave) {
            if (ISS) wait_units_in_flight(X32_D - 2);
            mf_ring_barrier();
        }
        if (ASMRD) {
#pragma unroll
            for (int f = 0; f < PF; f++) asm volatile("" : "+v"(afr[f]));
        }
    };
    i32x16_t acc_a, acc_b;
#pragma unroll
    for (int i = 0; i < 16; i++) acc_a[i] = 0, acc_b[i] = 0;
    // one iteration past the last tile: its MFMAs run on the clamped refill (discarded), its screening is the last tile's
    // (NODEF: nothing left to screen there, the iteration only keeps the barrier counts of the waves equal)
    if (NODEF) {
        if (issuer) {
            for (uint32_t it = 0; it <= my_tiles; it++) unit(std::true_type{}, acc_a, acc_a, it == my_tiles);
        } else {
            for (uint32_t it = 0; it <= my_tiles; it++) unit(std::false_type{}, acc_a, acc_a, it == my_tiles);
        }
    } else if (issuer) {
        for (uint32_t it = 0;;) {
            unit(std::true_type{}, acc_a, acc_b, it == my_tiles);
            if (it++ == my_tiles) break;
            unit(std::true_type{}, acc_b, acc_a, it == my_tiles);
            if (it++ == my_tiles) break;
        }
    } else {
        for (uint32_t it = 0;;) {
            unit(std::false_type{}, acc_a, acc_b, it == my_tiles);
            if (it++ == my_tiles) break;
            unit(std::false_type{}, acc_b, acc_a, it == my_tiles);
            if (it++ == my_tiles) break;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (wq_n) flush_wave_queue();
    if (COUNT && lane == 0 && blockIdx.x < 4) printf("x32l wg %u wave %d: exact test in %u of %u units, %u groups\n", blockIdx.x, wave, n_fired, n_units, n_groups);
}
#pragma clang diagnostic pop


}  // namespace vsg
