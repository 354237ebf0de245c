// mfma_i8ks_kernels.hpp -- K-split MFMA filter for int8 / uint8 rows of 1 KiB against 256 queries (BASELINE config 3).
//
// Config 3 is balanced three ways: at 8 TB/s a CU receives 32 rows x 1 KiB every ~2000 cycles, the int8 matrix pipes need
// 2048 cycles for those rows x 256 queries, and with 16 queries per wave (k_mfma_filter_lowp<16 waves, NQW = 1>) the 16
// waves pull 512 KiB of A fragments out of LDS per 32 rows, which is 2048 cycles of the 256 B/clk LDS as well.  Three
// units at 100 % cannot overlap perfectly; measured: matrix pipe 43-46 % busy (profiles/r01_lowp_pmc.txt).
//
// This kernel halves the LDS traffic without giving up the 4 waves per SIMD that hide the ds_read -> MFMA latency:
//   * wave w = (g, kh): query group g = w & 7 (32 queries = two 16-column blocks), k-half kh = w >> 3 (k-steps 8 kh .. 8 kh + 7);
//     its query fragments are 2 x 8 x 4 = 64 VGPRs, as before.  Every A fragment (16 rows x 64 B) read from LDS now feeds
//     TWO MFMAs, and only the 8 waves of its k-half read it: 256 KiB of fragment reads per 32 rows instead of 512;
//   * the two k-halves of a dot product meet through LDS: wave (g, kh) keeps the rows of M-block kh and sends its partial
//     sums for the other block to its partner (g, 1 - kh): 2 KiB out and 2 KiB in per wave and tile (64 KiB per tile);
//   * the exchange is deferred by one tile, so it needs no barrier of its own: partials written before the ring barrier
//     of the next tile are read right after it, and the screening of tile t-1 rides inside the MFMA stream of tile t.
//     A slot is rewritten one whole tile after the partner read it; a per-wave "consumed" counter makes that formal;
//   * int8 Cosine: a lane first tests the largest of its 8 dots against the smallest threshold of its 4 rows x 2 queries
//     (one cvt, two multiplies, two compares); the per-value screen and the exact IEEE score (IP.cpp:264-271) run only
//     in the handful of waves where that fires.
// Same ring (3 x 32 KiB of whole rows, two units in flight, counted vmcnt, XOR-swizzled image), emission queue and
// scores as k_mfma_filter_lowp; results are identical.
#pragma once
#include "mfma_lowp_kernels.hpp"

namespace vsg {

constexpr int KS_NW = 16, KS_RT = 32, KS_STAGE = 32768, KS_NS = 3, KS_D = 2, KS_NAUX = 4;
constexpr int KS_AUX_OFF = KS_NS * KS_STAGE;              // KS_NAUX x 256 B: per-tile aux values (32 rows x 4 B used)
constexpr int KS_XCHG_OFF = KS_AUX_OFF + KS_NAUX * 256;   // 16 waves x [2 column blocks][64 lanes][16 B]
constexpr int KS_FLAG_OFF = KS_XCHG_OFF + KS_NW * 2048;   // consumed[16], flush[2]
constexpr int KS_EQ_OFF = KS_FLAG_OFF + 128;
constexpr int KS_CONST_OFF = KS_EQ_OFF + MF_EQ_BYTES;     // 1024 threads x {tau[2], qaux[2]}: read back by the emitting path only
constexpr int KS_LDS_BYTES = KS_CONST_OFF + KS_NW * 64 * 16;
static_assert(KS_LDS_BYTES <= 160 * 1024, "LDS budget");

#ifndef KS_PF
#define KS_PF 2   // fragments in flight before the first MFMA: 4 spills query fragments at 128 VGPRs
#endif
#ifndef KS_EARLY
#define KS_EARLY 1
#endif

// MODE bits: 1 = s_setprio around the lo waves' MFMA block (DMODE 0 only); 2 = roles (DMODE 1, see below); 8 = diagnosis build (P.dbg: bit 0 skip screening, bit 1 skip fragment reads + MFMA, bit 2 skip the row DMA)
template <int LK, int MODE = 0>
__global__ __launch_bounds__(KS_NW * 64, 1) void k_i8_filter_ksplit(LowpParams P) {
    using Ops = LowpOps<LK>;
    constexpr bool SETPRIO = (MODE & 1) != 0, DIAG = (MODE & 8) != 0;
    constexpr int DMODE = (MODE >> 1) & 1;
    const int dbg = DIAG ? P.dbg : 0;
    static_assert(LK == LP_I8 || LK == LP_U8, "integer rows");
    constexpr int KH = 8;       // k-steps per wave
    constexpr int IPW = 2;      // DMA pieces (one whole row each) per wave and unit
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave & 7, kh = wave >> 3;
    const int m16 = lane & 15, kq = lane >> 4;
    const int qtile = (int)blockIdx.y;

    u32x4_t qf[2][KH];
    {
        const u32x4_t *src = reinterpret_cast<const u32x4_t *>(P.qfrag) + ((size_t)((qtile * 16 + g * 2) * 16 + KH * kh)) * 64 + lane;
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int s = 0; s < KH; s++) qf[nt][s] = src[(size_t)(nt * 16 + s) * 64];
    }
    const uint32_t lds0 = mf_lds_offset(lds);
    const uint32_t const_off = lds0 + KS_CONST_OFF + (uint32_t)tid * 16u;
    const int qidx0 = qtile * 256 + g * 32 + m16;   // column block nt: qidx0 + 16 nt
    // int8 Cosine screen (see k_mfma_filter_lowp): 1 - dot/(nx nq) <= tau  <=>  dot >= (1 - tau) nq nx; cosq carries a
    // margin for every rounding on either side.  cmin = the smaller of the lane's two cosq (NaN if either is).  tau and the
    // query aux values are parked in LDS: only the emitting path (a handful of waves per million rows) needs them again
    auto cosq_of = [](float tau_q, uint32_t qaux_q) -> float {
        const float omt = 1.0f - tau_q;
        return (omt - 1e-5f * (1.0f + fabsf(omt))) * __uint_as_float(qaux_q);
    };
    float cmin;
    {
        const uint32_t qa0 = P.qaux[qidx0], qa1 = P.qaux[qidx0 + 16];
        const float t0 = P.tau[qidx0], t1 = P.tau[qidx0 + 16];
        const u32x4_t cv = {__float_as_uint(t0), __float_as_uint(t1), qa0, qa1};
        asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(const_off), "v"(cv) : "memory");
        const float c0 = cosq_of(t0, qa0), c1 = cosq_of(t1, qa1);
        cmin = (c0 != c0 || c1 != c1) ? __builtin_nanf("") : fminf(c0, c1);
    }
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int s = 0; s < KH; s++) asm volatile("" : "+v"(qf[nt][s]));
    asm volatile("" : "+v"(cmin));

    // staging: piece t of this wave is row 2 wave + t of the unit, LDS bytes [1024 row, 1024 row + 1024)
    uint32_t voff[IPW];
#pragma unroll
    for (int t = 0; t < IPW; t++) {
        const uint32_t row = (uint32_t)(IPW * wave + t);
        voff[t] = row * P.row_stride + (uint32_t)(lane >> 4) * 256u + ((((uint32_t)lane & 15u) ^ (row & 15u)) * 16u);
    }
    const uint32_t lds_stage_wave_off = (uint32_t)(wave * IPW * 1024);
    const bool aux_loader = wave == 0;
    char *aux_lds = lds + KS_AUX_OFF;
    uint32_t *eq_n = reinterpret_cast<uint32_t *>(lds + KS_EQ_OFF);
    uint4 *eq = reinterpret_cast<uint4 *>(lds + KS_EQ_OFF + 16);
    uint32_t eq_n_off = lds0 + KS_EQ_OFF, eq_off = eq_n_off + 16;   // (not const: nested generic lambdas fail to capture const locals)
    const uint32_t aux_rd_off = lds0 + KS_AUX_OFF + (uint32_t)(kh * 64 + kq * 16);
    const uint32_t xw_off = lds0 + KS_XCHG_OFF + (uint32_t)wave * 2048u + (uint32_t)lane * 16u;         // my slot
    const uint32_t xr_off = lds0 + KS_XCHG_OFF + (uint32_t)(wave ^ 8) * 2048u + (uint32_t)lane * 16u;   // partner's slot
    const uint32_t my_flag_off = lds0 + KS_FLAG_OFF + (uint32_t)wave * 4u;          // exchanges of MY slot the partner has read
    const uint32_t pa_flag_off = lds0 + KS_FLAG_OFF + (uint32_t)(wave ^ 8) * 4u;
    const uint32_t flush_off = lds0 + KS_FLAG_OFF + 64u;
    if (tid < 32) {
        const uint32_t zero = 0, o = lds0 + KS_FLAG_OFF + (uint32_t)tid * 4u;
        asm volatile("ds_write_b32 %0, %1" ::"v"(o), "v"(zero) : "memory");
    }
    if (tid == 0) *eq_n = 0;

    // fragment addresses inside a slot: M-block (mtl ^ kh) -- block 0 of a wave is the one it keeps -- row m16, 256-B chunk
    // 2 kh + jj / 4, swizzled 16-B piece ((4 (jj % 4) + kq) ^ m16)
    // = fa[mtl] ^ (64 jm): 4 jm only touches bits 2-3 of the piece index, the row / chunk part has its low 8 bits clear
    uint32_t fa[2];
#pragma unroll
    for (int mtl = 0; mtl < 2; mtl++) fa[mtl] = (uint32_t)(((mtl ^ kh) * 16 + m16) * 1024 + 2 * kh * 256 + ((kq ^ m16) * 16));

    const uint32_t step = gridDim.x;
    auto tile_row0 = [&](uint32_t t) -> uint32_t { return (P.tile_first + t * P.tile_step) * KS_RT; };
    const char *rp_f[IPW];
    const uint32_t *ap_f;
    uint32_t cur_slab = 0xFFFFFFFFu;
    uint64_t cur_sbase = 0, cur_abase = 0;
    auto make_ptrs = [&](uint32_t t) {
        const uint32_t tt = t < P.n_tiles ? t : P.n_tiles - 1;
        const uint32_t r0 = __builtin_amdgcn_readfirstlane(tile_row0(tt));
        const uint32_t sidx = r0 >> P.slab_shift;
        if (sidx != cur_slab) {   // scalar loads by hand: left to hipcc they become vector loads + vmcnt(0) (see lowp kernel)
            cur_slab = sidx;
            const char *const *sp = P.slabs + sidx;
            const uint32_t *const *axp = P.aux_slabs + sidx;
            asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(cur_sbase), "=&s"(cur_abase)
                         : "s"(sp), "s"(axp)
                         : "memory");
        }
        const uint32_t in_slab = r0 & P.slab_mask;
        const char *tb = reinterpret_cast<const char *>(cur_sbase) + (size_t)in_slab * P.row_stride;
        if (r0 + KS_RT <= P.n_rows) {
#pragma unroll
            for (int i = 0; i < IPW; i++) rp_f[i] = tb + voff[i];
        } else {   // last, partial tile: rows past the end re-read the last row (their results are masked by nvalid)
#pragma unroll
            for (int i = 0; i < IPW; i++) {
                uint32_t row = (uint32_t)(IPW * wave + i);
                const uint32_t last = P.n_rows - 1 - r0;
                const uint32_t back = row > last ? (row - last) * P.row_stride : 0u;
                rp_f[i] = tb + voff[i] - back;
            }
        }
        uint32_t arow = r0 + lane;
        if (arow >= P.n_rows) arow = P.n_rows - 1;
        ap_f = reinterpret_cast<const uint32_t *>(cur_abase) + (arow & P.slab_mask);
    };
    auto issue = [&](uint32_t slot, uint32_t abuf_i) {
        const uint32_t base = slot * KS_STAGE + lds_stage_wave_off;
        if (!(dbg & 4)) {
#pragma unroll
            for (int i = 0; i < IPW; i++) glds16<2>(rp_f[i], base + i * 1024, lds);
        }
        if (aux_loader) glds4(ap_f, abuf_i * 256, aux_lds);
    };

    uint32_t tile = blockIdx.x;
    uint32_t ftile = tile, fbuf = 0;
    make_ptrs(ftile);
    issue(0, 0);
    ftile += step;
    make_ptrs(ftile);
    fbuf = 1;
    issue(1, 1);

    uint32_t slot_c = 0, abuf_prev = 0, n_done = 0;
    uint32_t prev_r0 = 0;
    i32x4_t own[2] = {i32x4_t{0, 0, 0, 0}, i32x4_t{0, 0, 0, 0}};

    // ---- screening + emission of one finished tile: v = the lane's 8 complete dots (rows 16 kh + 4 kq + i, columns m16 of the
    // wave's two column blocks), av = the 4 rows' aux values
    auto emit_tile = [&](const i32x4_t (&v)[2], const u32x4_t av, uint32_t r0, auto epi_tag) -> bool {
        constexpr int EPI = decltype(epi_tag)::value;
        const uint32_t nvalid = P.n_rows - r0;
        bool emitted = false;
        u32x4_t cv;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(cv) : "v"(const_off) : "memory");
        const float tau[2] = {__uint_as_float(cv[0]), __uint_as_float(cv[1])};
        const uint32_t qaux[2] = {cv[2], cv[3]};
        const float cosq[2] = {cosq_of(tau[0], qaux[0]), cosq_of(tau[1], qaux[1])};
        const int qidx[2] = {qidx0, qidx0 + 16};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t lrow = (uint32_t)(kh * 16 + kq * 4 + i);
            const uint32_t a = av[i];
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                const int dot = v[nt][i];
                if (EPI == LE_I8_COS) {
                    if ((float)dot < cosq[nt] * __uint_as_float(a)) continue;   // NaN thresholds fall through to the exact test
                }
                float sc;
                if (EPI == LE_I8_L2) sc = (float)((int)a + (int)qaux[nt] - 2 * dot);
                else if (EPI == LE_I8_IP) sc = (float)(1 - dot);
                else if (EPI == LE_U8_IP) sc = (float)(1 - (dot + 128 * (int)a + (int)qaux[nt]));
                else sc = __fsub_rn(1.0f, __fdiv_rn((float)dot, __fmul_rn(__uint_as_float(a), __uint_as_float(qaux[nt]))));
                if (lrow < nvalid && sc <= tau[nt]) {
                    const uint32_t row = r0 + lrow;
                    const uint32_t pos = mf_queue_reserve(eq_n_off);
                    if (pos < (uint32_t)MF_EQ_CAP) {
                        mf_queue_write(eq_off + pos * 16, row, (uint32_t)qidx[nt], __float_as_uint(sc));
                    } else {
                        uint32_t s = atomicAdd(&P.counts[qidx[nt]], 1u);
                        if (s < P.cap) P.cand[(size_t)qidx[nt] * P.cap + s] = make_uint2(row, __float_as_uint(sc));
                        emitted = true;
                    }
                }
            }
        }
        return emitted;
    };
    // cheap lane-level test: may any of the lane's 8 values pass?  (superset of the per-value tests of emit_tile)
    auto screen = [&](const i32x4_t (&v)[2], const u32x4_t av, auto epi_tag) -> bool {
        constexpr int EPI = decltype(epi_tag)::value;
        if (EPI == LE_I8_COS) {
            // dot >= c nx for some (row, column) implies  max dot >= min(cmin nmin, cmin nmax)  (nx >= 0, c >= cmin); NaN passes
            const float n0 = __uint_as_float(av[0]), n1 = __uint_as_float(av[1]), n2 = __uint_as_float(av[2]), n3 = __uint_as_float(av[3]);
            const float nmin = fminf(fminf(n0, n1), fminf(n2, n3)), nmax = fmaxf(fmaxf(n0, n1), fmaxf(n2, n3));
            int m = max(max(v[0][0], v[0][1]), max(v[0][2], v[0][3]));
            m = max(m, max(max(v[1][0], v[1][1]), max(v[1][2], v[1][3])));
            const float mf = (float)m;
            return !(mf < cmin * nmin) || !(mf < cmin * nmax);
        } else {
            bool any = false;
            u32x4_t cv;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(cv) : "v"(const_off) : "memory");
            const float tau[2] = {__uint_as_float(cv[0]), __uint_as_float(cv[1])};
            const uint32_t qaux[2] = {cv[2], cv[3]};
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    const int dot = v[nt][i];
                    const uint32_t a = av[i];
                    if (EPI == LE_I8_L2) any |= (float)((int)a + (int)qaux[nt] - 2 * dot) <= tau[nt];
                    else if (EPI == LE_I8_IP) any |= (float)(1 - dot) <= tau[nt];
                    else any |= (float)(1 - (dot + 128 * (int)a + (int)qaux[nt])) <= tau[nt];
                }
            return any;
        }
    };

    // Roles (DMODE 1).  The four waves of a SIMD (w, w+4, w+8, w+12) leave the ring barrier together and share one matrix
    // pipe, so their MFMA blocks execute one after the other whatever the code does; what the code can choose is who does
    // its bookkeeping (refill request, exchange read, screening) while somebody else's MFMAs run.  Waves 0-7 ("hi", the
    // first two of every SIMD, s_setprio 3 / 2) go straight from the barrier into their MFMA block and do the bookkeeping
    // afterwards; waves 8-15 ("lo", s_setprio 1 / 0) do the bookkeeping first -- under the hi waves' MFMAs -- and their
    // MFMA block last.  Exchange partners (w, w ^ 8) are always one hi and one lo wave.
    const bool hi = DMODE == 1 && wave < 8;
    if (DMODE == 1) {
        switch (wave >> 2) {
        case 0: asm volatile("s_setprio 3"); break;
        case 1: asm volatile("s_setprio 2"); break;
        case 2: asm volatile("s_setprio 1"); break;
        default: asm volatile("s_setprio 0"); break;
        }
    }
    auto run = [&](auto epi_tag) {
        uint32_t ff_cur = 0, ff_next = 0;   // queue flush decisions (wave 0 decides two tiles ahead: no LDS round trip in front of a barrier)
        for (; tile < P.n_tiles; tile += step) {
            // unit landed: one more unit (2 pieces, + its aux piece in wave 0) may stay in flight
            if (aux_loader) lowp_wait_vmcnt(IPW + 1);
            else lowp_wait_vmcnt(IPW);
            mf_ring_barrier();
            if (ff_cur) mf_flush_queue<KS_NW * 64>(eq_n, eq, P.counts, P.cand, P.cap);   // nobody is emitting here
            ff_cur = ff_next;

            auto request_ahead = [&]() {   // unit n+2 into the slot unit n-1 occupied
                uint32_t slot_p = slot_c + KS_D;
                if (slot_p >= KS_NS) slot_p -= KS_NS;
                ftile += step;
                make_ptrs(ftile);
                fbuf = (fbuf + 1) & (KS_NAUX - 1);
                issue(slot_p, fbuf);
            };
            u32x4_t px0 = {0, 0, 0, 0}, px1 = {0, 0, 0, 0}, av = {0, 0, 0, 0};
            uint32_t ffn = 0, fill = 0, seen = 0;
            // partner's partial sums and the aux values of the previous tile, the flush decision for the next tile
            // (first tile: the slots hold nothing yet; what is read is ignored below)
            auto read_exchange = [&]() {
                asm volatile("ds_read_b32 %0, %1" : "=v"(ffn) : "v"(flush_off + ((n_done + 1u) & 3u) * 4u) : "memory");
                const uint32_t eqn_addr = eq_n_off;   // (an asm operand alone does not capture the variable here)
                if (wave == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(fill) : "v"(eqn_addr) : "memory");
                asm volatile("ds_read_b128 %0, %1" : "=v"(px0) : "v"(xr_off) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(px1) : "v"(xr_off) : "memory");
                asm volatile("ds_read_b128 %0, %1" : "=v"(av) : "v"(aux_rd_off + abuf_prev * 256u) : "memory");
            };
            auto finish_prev_tile = [&]() {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ffn), "+v"(fill), "+v"(px0), "+v"(px1), "+v"(av)::"memory");
                if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(pa_flag_off), "v"(n_done) : "memory");
                if (wave == 0) {
                    const uint32_t f = fill >= (uint32_t)(MF_EQ_CAP / 2) ? 1u : 0u;
                    if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(flush_off + ((n_done + 2u) & 3u) * 4u), "v"(f) : "memory");
                }
                ff_next = __builtin_amdgcn_readfirstlane(ffn);
                i32x4_t v[2];
                v[0] = own[0] + __builtin_bit_cast(i32x4_t, px0);
                v[1] = own[1] + __builtin_bit_cast(i32x4_t, px1);
                if (!(dbg & 1)) {
                    bool hit = screen(v, av, epi_tag) && n_done > 0;
                    if (__any(hit)) {
                        hit = emit_tile(v, av, prev_r0, epi_tag);
                        if (__any(hit)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // atomics share the VM counter with the ring
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                }
            };

            typename Ops::acc_t acc[2][2];
#pragma unroll
            for (int mtl = 0; mtl < 2; mtl++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++) acc[mtl][nt] = typename Ops::acc_t{0, 0, 0, 0};
            const char *sbase = lds + slot_c * KS_STAGE;
            constexpr int NFRAG = 2 * KH;
            auto mfma_part = [&](auto f0_tag, auto f1_tag) {
                constexpr int F0 = decltype(f0_tag)::value, F1 = decltype(f1_tag)::value, N = F1 - F0;
                constexpr int PF = N < KS_PF ? N : KS_PF;
                u32x4_t afr[N];
#pragma unroll
                for (int f = F0; f < F1; f++) {
                    const int jj = f / 2, mtl = f % 2;
                    afr[f - F0] = *reinterpret_cast<const u32x4_t *>(sbase + (fa[mtl] ^ (uint32_t)((jj % 4) * 64)) + (jj / 4) * 256);
                    if (LK == LP_U8) afr[f - F0] ^= 0x80808080u;
                }
#pragma unroll
                for (int f = F0; f < F1; f++) {
                    const int jj = f / 2, mtl = f % 2;
#pragma unroll
                    for (int nt = 0; nt < 2; nt++) acc[mtl][nt] = Ops::mma(afr[f - F0], qf[nt][jj], acc[mtl][nt]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
                for (int f = 0; f < N - PF; f++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, PF * 2, 0);
            };
            using I0 = std::integral_constant<int, 0>;
            using I8 = std::integral_constant<int, 8>;
            using I16 = std::integral_constant<int, NFRAG>;
            // (the "+v" uses: otherwise the MFMA chains sink below whatever follows, behind its lgkmcnt(0))
            if (hi) {
                // has the partner read my slot?  It did so at the top of its previous MFMA block; the answer is back long
                // before it is needed
                asm volatile("ds_read_b32 %0, %1" : "=v"(seen) : "v"(my_flag_off) : "memory");
                if (!(dbg & 2)) mfma_part(I0{}, I16{});
                asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(seen));
                request_ahead();
                read_exchange();
                finish_prev_tile();
            } else {
                read_exchange();
                request_ahead();
                finish_prev_tile();
                if (SETPRIO) asm volatile("s_setprio 1" ::: "memory");
                if (!(dbg & 2)) mfma_part(I0{}, I8{});
                asm volatile("ds_read_b32 %0, %1" : "=v"(seen) : "v"(my_flag_off) : "memory");
                if (!(dbg & 2)) mfma_part(I8{}, I16{});
                if (SETPRIO) asm volatile("s_setprio 0\n\ts_waitcnt lgkmcnt(0)" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(seen));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(seen));
            }
            // my slot is free once the partner has read exchange n_done - 1 from it
            while (__builtin_amdgcn_readfirstlane(seen) < n_done)
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(my_flag_off) : "memory");
            {
                const u32x4_t s0 = __builtin_bit_cast(u32x4_t, acc[1][0]), s1 = __builtin_bit_cast(u32x4_t, acc[1][1]);
                asm volatile("ds_write_b128 %0, %1" ::"v"(xw_off), "v"(s0) : "memory");
                asm volatile("ds_write_b128 %0, %1 offset:1024" ::"v"(xw_off), "v"(s1) : "memory");
            }
            own[0] = acc[0][0];
            own[1] = acc[0][1];
            prev_r0 = tile_row0(tile);
            abuf_prev = n_done & (KS_NAUX - 1);
            slot_c = slot_c + 1 == KS_NS ? 0 : slot_c + 1;
            n_done++;
        }
        // last tile of this workgroup
        mf_ring_barrier();
        if (ff_cur) mf_flush_queue<KS_NW * 64>(eq_n, eq, P.counts, P.cand, P.cap);
        if (n_done > 0) {
            u32x4_t px0, px1, av;
            asm volatile("ds_read_b128 %0, %1" : "=v"(px0) : "v"(xr_off) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(px1) : "v"(xr_off) : "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(av) : "v"(aux_rd_off + abuf_prev * 256u) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(px0), "+v"(px1), "+v"(av)::"memory");
            i32x4_t v[2];
            v[0] = own[0] + __builtin_bit_cast(i32x4_t, px0);
            v[1] = own[1] + __builtin_bit_cast(i32x4_t, px1);
            if (__any(screen(v, av, epi_tag))) (void)emit_tile(v, av, prev_r0, epi_tag);
        }
    };
    if (LK == LP_U8) {
        if (P.epi == LE_U8_IP) run(std::integral_constant<int, LE_U8_IP>{});
        else run(std::integral_constant<int, LE_I8_L2>{});
    } else {
        if (P.epi == LE_I8_COS) run(std::integral_constant<int, LE_I8_COS>{});
        else if (P.epi == LE_I8_L2) run(std::integral_constant<int, LE_I8_L2>{});
        else run(std::integral_constant<int, LE_I8_IP>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    mf_flush_queue<KS_NW * 64>(eq_n, eq, P.counts, P.cand, P.cap);
}

}  // namespace vsg
