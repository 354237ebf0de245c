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

constexpr int X32_RT = 32;            // rows per unit (= per tile)
constexpr int X32_ROWB = 1040;        // LDS bytes per row image: 1 KiB + 16, so that 32 rows x one 16-byte chunk (ds_read_b128 in
                                      // lane groups of 16) fall on 16 distinct 16-byte bank slots -- no swizzle, one address register
constexpr int X32_UNIT = X32_RT * X32_ROWB;   // bytes per ring slot: 32 whole row images
constexpr int X32_NW = 8;             // waves
constexpr int X32_KS = 32;            // k-steps of 32 bytes
constexpr int X32_AUXB = 8;           // per-unit aux buffers (256 B each; a unit's values live from its request to its screening)
constexpr int X32_QT = 256;           // queries per workgroup
constexpr int X32_WQ_CAP = 64;         // records per wave-private candidate queue (one ballot's worth always fits)
constexpr int x32_lds_bytes(int ns) { return ns * X32_UNIT + X32_AUXB * 256 + X32_NW * X32_WQ_CAP * 16; }

// VAR bits (the shipped build instantiates 32768 | 1 only; the rest is the tuning build's, profiles/r03_c3_x32.txt):
// 1 = EARLY (rows land one barrier early, fragments prefetched across the barrier), 2 = requests spread over the stream
// instead of behind the first MFMAs, 8 = two accumulator chains (even / odd k-steps), 16 = s_setprio 1 for the
// second-dispatched half of the waves, 16384 = inline-asm fragment reads with counted waits, 32768 = only waves 0-3 request rows
// X32_NS ring slots, X32_D units requested ahead
template <int LK, int EPI, int VAR, int X32_NS = 4, int X32_D = 3>
__global__ __launch_bounds__(X32_NW * 64, 2) void k_i8_filter_x32(LowpParams P) {
    static_assert(X32_D >= 1 && X32_D < X32_NS && (!((VAR & 1) != 0) || X32_D >= 2), "ring geometry");
    static_assert(LK == LP_I8 || LK == LP_U8, "int8 / uint8 rows with 4-byte aux values");
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
    wait_units_in_flight(EARLY ? X32_D - 2 : X32_D - 1);
    mf_ring_barrier();

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
                    if (f < X32_KS) afr[ks % PF] = read_frag(cslot, f);
                    else if (EARLY) afr[ks % PF] = read_frag(nslot, f - X32_KS);
                }
                if (ks >= PBASE && (ks - PBASE) % PSTEP == 0 && (ks - PBASE) / PSTEP < IPWX) issue_piece((ks - PBASE) / PSTEP);
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
        stamp(5);
        wait_units_in_flight(EARLY ? X32_D - 2 : X32_D - 1);
        stamp(2);
        mf_ring_barrier();
        stamp(3);
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

}  // namespace vsg
