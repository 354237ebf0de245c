// mfma_free_kernels.hpp -- barrier-free variant of the low-precision MFMA filter (mfma_lowp_kernels.hpp).
//
// k_mfma_filter_lowp synchronises its one resident workgroup with an s_barrier per ring unit, so all waves wait,
// request, read, multiply and screen in lock step (profiles/r01_lowp_pmc.txt: matrix pipe 43 % busy, waves parked
// 51 % of their cycles).  Here the barrier is replaced by two monotonic counters per ring slot, kept in LDS:
//
//   landed[s]    += 1 by every wave once ITS DMA pieces of a unit have reached slot s (its own vmcnt says so);
//                   unit g (slot g % NS, generation g / NS) may be read when landed[s] has NWAVES * (gen + 1) signals
//   released[s]  += 1 by every wave behind its last LDS read of the unit in slot s (the LDS executes a wave's
//                   operations in order, so the add is performed after the reads);
//                   unit h may be requested into slot s when released[s] has NWAVES * (h / NS) signals
//
// A wave signals "landed" for unit g+L at the top of unit g (L >= 1 units early), so nobody waits for a
// wave that merely has not ARRIVED at a unit yet: waves may run up to min(L, NS-D) units apart, and the four
// waves of a SIMD drift out of phase -- one screens or waits while another feeds the matrix pipe.
// Why the counts cannot be fooled: a wave signals generation G+1 of a slot only after it has passed the poll for
// generation G of that slot (L < NS), which needed every wave's generation-G signal; same for released (a wave reads
// generation G+1 only after the refill, which needed every wave's generation-G release).
// Survivors go to a wave-private LDS queue that the wave drains itself (no collective flush).  FILTER mode only.
#pragma once
#include "mfma_lowp_kernels.hpp"

namespace vsg {

constexpr int FREE_WQ_CAP = 32;  // records per wave-private queue
constexpr int free_lds_bytes(int nwaves, int ksteps, int rt, int ns, int stage, int d) {
    const int kch = ksteps / ((stage / rt) / 64);
    const int naux = (kch - 1 + d) / kch + 1;
    return ns * stage + nwaves * 256 * naux + nwaves * (16 + FREE_WQ_CAP * 16) + 128;
}

template <int LK, int KSTEPS, int RT, int NWAVES, int NQW, int NS, int STAGE, int D, int L>
__global__ __launch_bounds__(NWAVES * 64, 1) void k_mfma_filter_free(LowpParams P) {
    using Ops = LowpOps<LK>;
    using acc_t = typename Ops::acc_t;
    constexpr int MT = RT / 16;
    constexpr int SEG = STAGE / RT;
    constexpr int KSUB = SEG / 64;
    static_assert(KSTEPS % KSUB == 0, "row bytes must be a multiple of the stage segment");
    constexpr int KCH = KSTEPS / KSUB;
    constexpr int IPW = (STAGE / 1024) / NWAVES;
    static_assert(IPW >= 1 && SEG * RT == STAGE && KSUB >= 1, "stage geometry");
    static_assert(L >= 1 && L <= D - 1 && D <= NS - 1, "look-ahead geometry");
    constexpr int TA = (KCH - 1 + D) / KCH;
    constexpr int NAUX = TA + 1;
    static_assert((D - 1) * IPW + D <= 40, "vmcnt immediate table");
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15;
    const int kq = lane >> 4;
    const int qtile = (int)blockIdx.y;

    u32x4_t qf[NQW][KSTEPS];
    {
        const u32x4_t *src = reinterpret_cast<const u32x4_t *>(P.qfrag) + ((size_t)((qtile * NWAVES + wave) * NQW) * KSTEPS) * 64 + lane;
#pragma unroll
        for (int nt = 0; nt < NQW; nt++)
#pragma unroll
            for (int s = 0; s < KSTEPS; s++) qf[nt][s] = src[(size_t)(nt * KSTEPS + s) * 64];
    }
    int qidx[NQW];
    uint32_t qaux[NQW];
    float tau[NQW];
#pragma unroll
    for (int nt = 0; nt < NQW; nt++) {
        qidx[nt] = qtile * (NWAVES * 16 * NQW) + wave * (16 * NQW) + nt * 16 + m16;
        qaux[nt] = P.qaux[qidx[nt]];
        tau[nt] = P.tau[qidx[nt]];
    }
#pragma unroll
    for (int nt = 0; nt < NQW; nt++) {
#pragma unroll
        for (int s = 0; s < KSTEPS; s++) asm volatile("" : "+v"(qf[nt][s]));
        asm volatile("" : "+v"(qaux[nt]), "+v"(tau[nt]));
    }
    float cosq[NQW];  // int8 Cosine screen, see k_mfma_filter_lowp
#pragma unroll
    for (int nt = 0; nt < NQW; nt++) {
        const float omt = 1.0f - tau[nt];
        cosq[nt] = (omt - 1e-5f * (1.0f + fabsf(omt))) * __uint_as_float(qaux[nt]);
    }

    uint32_t st_row[IPW], st_off[IPW];
#pragma unroll
    for (int t = 0; t < IPW; t++) {
        const uint32_t Lb = 1024u * (uint32_t)(IPW * wave + t) + 16u * (uint32_t)lane;
        const uint32_t row = Lb / SEG, slot = (Lb % SEG) / 16;
        st_row[t] = row;
        st_off[t] = (slot / 16) * 256 + (((slot % 16) ^ (row & 15)) * 16);
    }
    const uint32_t lds_stage_wave_off = (uint32_t)(wave * IPW * 1024);
    char *aux_lds = lds + NS * STAGE + wave * (256 * NAUX);
    const uint32_t aux_lds_off = mf_lds_offset(aux_lds);
    char *wq_base = lds + NS * STAGE + NWAVES * 256 * NAUX;
    const uint32_t wq_cnt_off = mf_lds_offset(wq_base) + (uint32_t)wave * (16 + FREE_WQ_CAP * 16);
    const uint32_t cnt_off = mf_lds_offset(wq_base + NWAVES * (16 + FREE_WQ_CAP * 16));  // landed[NS] then released[NS]
    static_assert(2 * NS * 4 <= 128, "counter block");
    {
        const uint32_t zero = 0;
        if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(wq_cnt_off), "v"(zero) : "memory");
        if (tid < 2 * NS) asm volatile("ds_write_b32 %0, %1" ::"v"(cnt_off + 4u * (uint32_t)tid), "v"(zero) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the only barrier of the kernel
    }
    // (one lane only: 64 lanes adding to one address would hold the LDS for 64 cycles per signal)
    constexpr uint32_t SIG = NWAVES;
    auto lds_add1 = [&](uint32_t off) {
        const uint32_t one = 1;
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(off), "v"(one) : "memory");
        else asm volatile("" ::: "memory");
    };
    // wait until landed[sa] has reached ta and released[sb] has reached tb (wrap-safe compares)
    auto poll2 = [&](uint32_t sa, uint32_t ta, uint32_t sb, uint32_t tb) {
        const uint32_t oa = cnt_off + 4u * sa, ob = cnt_off + 4u * (uint32_t)NS + 4u * sb;
        // (the spin bound only keeps a protocol bug from hanging the GPU: a legitimate wait is microseconds)
        for (uint32_t spins = 0; spins < (1u << 16); spins++) {
            uint32_t a, b;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(oa), "v"(ob) : "memory");
            a = __builtin_amdgcn_readfirstlane(a);
            b = __builtin_amdgcn_readfirstlane(b);
            if ((int)(a - ta) >= 0 && (int)(b - tb) >= 0) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    auto drain_wave_queue = [&]() {
        uint32_t n;
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(n) : "v"(wq_cnt_off) : "memory");
        n = min(n, (uint32_t)FREE_WQ_CAP);
        if ((uint32_t)lane < n) {
            u32x4_t r;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(wq_cnt_off + 16u + (uint32_t)lane * 16u) : "memory");
            const uint32_t s = atomicAdd(&P.counts[r[1]], 1u);
            if (s < P.cap) P.cand[(size_t)r[1] * P.cap + s] = make_uint2(r[0], r[2]);
        }
        const uint32_t zero = 0;
        if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(wq_cnt_off), "v"(zero) : "memory");
        // stores/atomics share the VM counter with the staged loads: drain so the counted waits stay exact
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };

    const uint32_t step = gridDim.x;
    auto tile_row0 = [&](uint32_t t) -> uint32_t { return (P.tile_first + t * P.tile_step) * RT; };
    const char *rp_f[IPW];
    const uint32_t *ap_f;
    uint32_t cur_slab = 0xFFFFFFFFu;
    uint64_t cur_sbase = 0, cur_abase = 0;
    auto make_ptrs = [&](uint32_t t, const char *(&rp)[IPW], const uint32_t *&ap) {
        uint32_t tt = t < P.n_tiles ? t : P.n_tiles - 1;
        const uint32_t r0 = tile_row0(tt);
        const uint32_t sidx = __builtin_amdgcn_readfirstlane(r0 >> P.slab_shift);
        if (sidx != cur_slab) {
            cur_slab = sidx;
            const char *const *sp = P.slabs + sidx;
            const uint32_t *const *axp = P.aux_slabs + sidx;
            asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(cur_sbase), "=&s"(cur_abase)
                         : "s"(sp), "s"(axp)
                         : "memory");
        }
        const char *sbase = reinterpret_cast<const char *>(cur_sbase);
        const uint32_t *abase = reinterpret_cast<const uint32_t *>(cur_abase);
#pragma unroll
        for (int i = 0; i < IPW; i++) {
            uint32_t row = r0 + st_row[i];
            if (row >= P.n_rows) row = P.n_rows - 1;
            rp[i] = sbase + (size_t)(row & P.slab_mask) * P.row_stride + st_off[i];
        }
        uint32_t arow = r0 + lane;
        if (arow >= P.n_rows) arow = P.n_rows - 1;
        ap = abase + (arow & P.slab_mask);
    };
    auto issue = [&](const char *const (&rpt)[IPW], const uint32_t *apt, int kc, uint32_t slot, bool with_aux, uint32_t abuf_i) {
        const uint32_t base = slot * STAGE + lds_stage_wave_off;
#pragma unroll
        for (int i = 0; i < IPW; i++) glds16<2>(rpt[i] + (size_t)kc * SEG, base + i * 1024, lds);
        if (with_aux) glds4(apt, abuf_i * 256, aux_lds);
    };

    uint32_t tile = blockIdx.x;
    uint32_t ftile = tile, fbuf = 0;
    make_ptrs(ftile, rp_f, ap_f);
    auto advance_frontier = [&]() {
        ftile += step;
        make_ptrs(ftile, rp_f, ap_f);
        fbuf = fbuf + 1 == NAUX ? 0 : fbuf + 1;
    };
#pragma unroll
    for (int u = 0; u < D; u++) {
        if (u > 0 && u % KCH == 0) advance_frontier();
        issue(rp_f, ap_f, u % KCH, u, (u % KCH) == 0, fbuf);
    }
    // units 0 .. L-1 have no earlier unit to be signalled from
#pragma unroll
    for (int j = 0; j < L; j++) {
        int n_out = (D - 1 - j) * IPW;
#pragma unroll
        for (int u = j + 1; u < D; u++) n_out += (u % KCH == 0) ? 1 : 0;
        lowp_wait_vmcnt(n_out);
        lds_add1(cnt_off + 4u * (uint32_t)j);
    }
    uint32_t slot_c = 0, gen_c = 0;                 // unit being consumed
    uint32_t slot_l = L;                            // unit g+L (signalled)
    uint32_t slot_p = D, gen_p = 0;                 // unit g+D (requested)
    uint32_t abuf = 0;
    bool pre_ok = false;

    for (; tile < P.n_tiles; tile += step) {
        acc_t acc[MT][NQW];
        u32x4_t auxv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NQW; nt++) acc[mt][nt] = acc_t{0, 0, 0, 0};

#pragma unroll
        for (int c = 0; c < KCH; c++) {
            {   // my pieces of unit c+L have landed; units c+L+1 .. c+D-1 may stay in flight
                int n_out = (D - 1 - L) * IPW;
#pragma unroll
                for (int j = L + 1; j < D; j++) n_out += ((c + j) % KCH == 0) ? 1 : 0;
                lowp_wait_vmcnt(n_out);
            }
            lds_add1(cnt_off + 4u * slot_l);
            // the counters were sampled during the previous unit (below); only a failed sample pays the LDS round
            // trip here -- through an LDS busy with 16 waves' fragment reads that is hundreds of cycles
            if (!pre_ok) poll2(slot_c, SIG * (gen_c + 1), slot_p, SIG * gen_p);
            {   // unit c+D into the slot every wave has released
                const int kc = (c + D) % KCH;
                if (kc == 0) advance_frontier();
                issue(rp_f, ap_f, kc, slot_p, kc == 0, fbuf);
            }
            const char *sbase = lds + slot_c * STAGE;
            constexpr int NFRAG = KSUB * MT;
            constexpr int PF = NFRAG < LOWP_PF ? NFRAG : LOWP_PF;
            {
                u32x4_t afr[NFRAG];
#pragma unroll
                for (int f = 0; f < NFRAG; f++) {
                    const int j = f / MT, mt = f % MT;
                    const char *rowp = sbase + (mt * 16 + m16) * SEG + (j / 4) * 256;
                    const int p = (4 * (j % 4) + kq) ^ m16;
                    afr[f] = *reinterpret_cast<const u32x4_t *>(rowp + p * 16);
                    if (LK == LP_U8) afr[f] ^= 0x80808080u;
                }
#pragma unroll
                for (int f = 0; f < NFRAG; f++) {
                    const int j = f / MT, mt = f % MT;
#pragma unroll
                    for (int nt = 0; nt < NQW; nt++) acc[mt][nt] = Ops::mma(afr[f], qf[nt][c * KSUB + j], acc[mt][nt]);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
#pragma unroll
                for (int f = 0; f < NFRAG - PF; f++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, NQW, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, PF * NQW, 0);
            }
            lds_add1(cnt_off + 4u * (uint32_t)NS + 4u * slot_c);  // behind this wave's last read of the slot
            {   // sample the next unit's two conditions now; the answer is back long before the MFMAs have drained
                const uint32_t sc1 = slot_c + 1 == NS ? 0 : slot_c + 1, sp1 = slot_p + 1 == NS ? 0 : slot_p + 1;
                const uint32_t ta = SIG * (gen_c + (sc1 == 0 ? 1u : 0u) + 1u), tb = SIG * (gen_p + (sp1 == 0 ? 1u : 0u));
                uint32_t a, b;
                asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(a), "=&v"(b)
                             : "v"(cnt_off + 4u * sc1), "v"(cnt_off + 4u * (uint32_t)NS + 4u * sp1)
                             : "memory");
                a = __builtin_amdgcn_readfirstlane(a);
                b = __builtin_amdgcn_readfirstlane(b);
                pre_ok = (int)(a - ta) >= 0 && (int)(b - tb) >= 0;
            }
            if (c == KCH - 1) {
                const uint32_t aoff = aux_lds_off + abuf * 256 + kq * 16;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(auxv[mt]) : "v"(aoff), "n"(mt * 64));
            }
            slot_c = slot_c + 1 == NS ? 0 : slot_c + 1;
            gen_c += slot_c == 0 ? 1 : 0;
            slot_l = slot_l + 1 == NS ? 0 : slot_l + 1;
            slot_p = slot_p + 1 == NS ? 0 : slot_p + 1;
            gen_p += slot_p == 0 ? 1 : 0;
        }

        // ---- epilogue (as k_mfma_filter_lowp, wave-private queue) ----
        const uint32_t r0 = tile_row0(tile);
        const uint32_t nvalid = P.n_rows - r0;
        bool emitted = false;
        if (MT == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(auxv[0]));
        else if (MT == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(auxv[0]), "+v"(auxv[1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(auxv[0]), "+v"(auxv[1]), "+v"(auxv[2]), "+v"(auxv[3]));
        auto epilogue = [&](auto epi_tag) {
            constexpr int EPI = decltype(epi_tag)::value;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t lrow = mt * 16 + kq * 4 + i;
                    const uint32_t av = auxv[mt][i];
#pragma unroll
                    for (int nt = 0; nt < NQW; nt++) {
                        float low;
                        if (LK == LP_I8 || LK == LP_U8) {
                            const int dot = (int)acc[mt][nt][i];
                            if (EPI == LE_I8_COS) {
                                if ((float)dot < cosq[nt] * __uint_as_float(av)) continue;
                            }
                            if (EPI == LE_I8_L2) low = (float)((int)av + (int)qaux[nt] - 2 * dot);
                            else if (EPI == LE_I8_IP) low = (float)(1 - dot);
                            else if (EPI == LE_U8_IP) low = (float)(1 - (dot + 128 * (int)av + (int)qaux[nt]));
                            else low = __fsub_rn(1.0f, __fdiv_rn((float)dot, __fmul_rn(__uint_as_float(av), __uint_as_float(qaux[nt]))));
                        } else {
                            const float dot = (float)acc[mt][nt][i];
                            const float ssum = __uint_as_float(av) + __uint_as_float(qaux[nt]);
                            const float a = (EPI == LE_FP_L2) ? (ssum - 2.0f * dot) : (1.0f - dot);
                            low = a - (P.cE * ssum + P.absE);
                        }
                        if (lrow < nvalid && ((LK == LP_I8 || LK == LP_U8) ? (low <= tau[nt]) : !(low > tau[nt]))) {
                            const uint32_t row = r0 + lrow;
                            const uint32_t pos = mf_queue_reserve(wq_cnt_off);
                            emitted = true;
                            if (pos < (uint32_t)FREE_WQ_CAP) {
                                mf_queue_write(wq_cnt_off + 16 + pos * 16, row, (uint32_t)qidx[nt], __float_as_uint(low));
                            } else {
                                uint32_t s = atomicAdd(&P.counts[qidx[nt]], 1u);
                                if (s < P.cap) P.cand[(size_t)qidx[nt] * P.cap + s] = make_uint2(row, __float_as_uint(low));
                            }
                        }
                    }
                }
            }
        };
        if (LK == LP_U8) {
            if (P.epi == LE_U8_IP) epilogue(std::integral_constant<int, LE_U8_IP>{});
            else epilogue(std::integral_constant<int, LE_I8_L2>{});
        } else if (LK == LP_I8) {
            if (P.epi == LE_I8_COS) epilogue(std::integral_constant<int, LE_I8_COS>{});
            else if (P.epi == LE_I8_L2) epilogue(std::integral_constant<int, LE_I8_L2>{});
            else epilogue(std::integral_constant<int, LE_I8_IP>{});
        } else {
            if (P.epi == LE_FP_L2) epilogue(std::integral_constant<int, LE_FP_L2>{});
            else epilogue(std::integral_constant<int, LE_FP_IP>{});
        }
        if (__any(emitted)) drain_wave_queue();
        abuf = abuf + 1 == NAUX ? 0 : abuf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}  // namespace vsg
