// mfma_wide_kernels.hpp -- the MFMA filter for rows wider than the other filters' registers hold: fp32 3072 < dim <= 8192
// (EK = 0), bf16 / fp16 2048 < dim <= 8192 (EK = 1 / 2: the stored elements are the MFMA operands, a stage holds 512 of them),
// int8 / uint8 4096 < dim <= 16384 (EK = 3 / 4: v_mfma_i32_16x16x64_i8 on the stored bytes, a stage holds 1024 of them; the
// int32 dot is exact, so the epilogue applies the reference's own score -- IP/...VNNI_INT8.h:11-76 -- and nothing is re-ranked;
// EK = 5: uint8 Cosine, whose score needs TWO per-row values -- the stored float norm and sum (x - 128) for the re-centred dot,
// IP_AVX512F_BW_VL_VNNI_UINT8.h:12-106 -- in the 16-byte aux records of the narrower uint8 Cosine path, k_row_aux_u8c).
//
// k_mfma_filter keeps the bf16 fragments of 64 queries in registers (16 per wave); at dim 4096 those alone are the whole
// register file of a CU.  This variant keeps 16 queries per column block of a WORKGROUP (one to four blocks) and splits the row's k
// range over the four waves: of every ring stage (16 rows x 1 KiB = 256 fp32 elements of k) wave w multiplies the quarter it
// requested itself (256 bytes of each row), so a wave holds KSTEPS / 4 fragments per block -- 128 registers at dim 4096, 256 at dim
// 8192 --, and the four partial dot products of a tile meet in LDS before the epilogue, which the waves share by column block.
// Swizzle, counted vmcnt, candidate queue and the bound E are k_mfma_filter's (mfma_kernels.hpp).  A batch of 64 queries is four query tiles (blockIdx.y): the rows are
// requested with the default cache policy, not non-temporal, so that the other three tiles' reads of a row hit L2 / the
// Infinity Cache (workgroups x, x + gridDim.x, ... share an XCD when gridDim.x is a multiple of 8).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mfma_kernels.hpp"

namespace vsg {

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
constexpr int MFW_QTILE = 16;               // queries per column block; a workgroup holds NQ of them (1 or 2)
constexpr int mfw_red_bytes(int nw) { return (nw - 1) * 64 * 16; }   // per column block: the other waves' partial sums
constexpr int mfw_norm_bytes(int ek) { return ek == 5 ? 3072 : 768; }   // three tiles of 64 aux values (4 B) / records (16 B)
constexpr int mfw_lds_bytes(bool probe, int nq = 1, int ns = 3, int ek = 0, int nw = 4) {   // (two tiles' partial sums)
    return ns * MF_STAGE_BYTES + mfw_norm_bytes(ek) + MF_EQ_BYTES + 2 * nq * mfw_red_bytes(nw) + (probe ? nq * MF_PM_TILES * 64 : 0);
}
// s_waitcnt needs an immediate: after unrolling, n is a constant and the switch folds to one instruction
__device__ static inline void mfw_wait_vmcnt(int n) {
    switch (n) {
#define VSG_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    VSG_W(0) VSG_W(1) VSG_W(2) VSG_W(3) VSG_W(4) VSG_W(5) VSG_W(6) VSG_W(7) VSG_W(8) VSG_W(9) VSG_W(10) VSG_W(11)
    VSG_W(12) VSG_W(13) VSG_W(14) VSG_W(15) VSG_W(16) VSG_W(17) VSG_W(18) VSG_W(19) VSG_W(20) VSG_W(21) VSG_W(22)
    VSG_W(23) VSG_W(24) VSG_W(25) VSG_W(26) VSG_W(27) VSG_W(28) VSG_W(29) VSG_W(30) VSG_W(31) VSG_W(32) VSG_W(33)
#undef VSG_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// NQ = 2: 32 queries per workgroup -- every fragment read from LDS feeds two MFMAs and a batch needs half the query tiles, i.e.
// half the passes over the rows (the tiles of a row tile share it through L2 only in part); 2 x KMINE fragments per wave.
// NS = ring slots.  Round 3 measured 6 slots no faster than 3 -- but then one wave worked per unit; since every wave works in every
// unit (round 4) the workgroup that is alone on its CU is bound by the bytes it has in flight: 5 slots (4 stages = 64 KiB requested
// ahead) are 4-21 % faster than 3 at every width and kind (profiles/r05_wide_ring_depth.txt, _depth2.txt).
//
// Round 5, late: NO BARRIER IN THE K LOOP.  A lone workgroup of four column blocks ran at 31.5 GB/s whatever the memory system had to
// spare (profiles/r05_wide_workgroup_rate.txt: 32, 64 or 128 workgroups on the chip, the same rate each) -- half of its unit period was
// not MFMA time but the ring barrier, the LDS latency behind it, and wave 0's epilogue with three waves waiting.  Now wave w owns a
// QUARTER OF EVERY ROW: bytes [256 w, 256 w + 256) of each row's 1 KiB stage segment -- its KPW consecutive k-steps -- are requested by
// wave w (four LDS-DMA instructions of 4 rows x 256 B), land in wave w's 4 KiB of the ring slot and are read by wave w alone.  What a
// wave waits for is its own vmcnt; a slot's refill follows the wave's own reads of it; the waves meet once per row tile, where the
// partial dot products are exchanged, and the epilogue is dealt over the waves by column block (wave nt owns block nt).
// NW = 8 (tuning only): EIGHT waves, an eighth of every row each (128 B per stage, two requests of 8 rows x 128 B).  With four waves
// of 512 registers a SIMD has one wave, and whatever that wave waits for -- above all the issue of its LDS-DMA requests, 60-185 cycles
// apiece -- the matrix pipe waits for too (a lone workgroup of four column blocks: 43 GB/s, a third of it MFMA time); eight waves would
// give every SIMD a second wave to multiply meanwhile.  Built and measured for four column blocks at 96 k-steps: bit-identical replies,
// but 192 registers of fragments + 16 accumulators + ~90 others do not fit a wave's 256 -- 50-80 spilled, some inside the k loop --
// and the filter ran at HALF the rate (bf16 3072, batch 128: 1.77 against 4.8 TB/s; profiles/r05_wide_eight_waves.txt).  Two column
// blocks per workgroup would fit, and then a batch crosses the rows twice as often: not built.
template <int KSTEPS, int MODE, int AUX = 0, int EK = 0, int NQ = 1, int NS = 3, int NW = 4>
__global__ __launch_bounds__(64 * NW, 1) void k_mfma_filter_wide(MfmaParams P) {
    constexpr int RT = 16;
    constexpr bool INT8 = EK >= 3;
    constexpr bool U8C = EK == 5;                      // uint8 Cosine: 16-byte aux records {norm, sum (x - 128), 0, 0}
    constexpr int NORM_BYTES = mfw_norm_bytes(EK), NORM_PAR = NORM_BYTES / 3;   // three tiles' aux values (see the tile-end barrier)
    constexpr int EB = EK == 0 ? 4 : (INT8 ? 1 : 2);   // bytes per stored element
    constexpr int KC = (MF_STAGE_BYTES / EB) / RT;   // 256 (fp32) / 512 (bf16, fp16) / 1024 (int8, uint8) elements per row per stage
    constexpr int SEG = KC * EB;                     // 1 KiB
    constexpr int KSUB = KC / (INT8 ? 64 : 32);      // 8 / 16 / 16 MFMA k-steps per stage
    typedef int i32x4w_t __attribute__((ext_vector_type(4)));
    using acc_v = typename std::conditional<INT8, i32x4w_t, f32x4_t>::type;
    static_assert(KSTEPS % KSUB == 0, "the row is a whole number of stages");
    constexpr int KCH = KSTEPS / KSUB;               // stages per row tile
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    constexpr int KPW = KSUB / NW;                   // k-steps per wave per stage: 2 (fp32) or 4 with four waves -- BW bytes of the row
    static_assert(KSUB % NW == 0, "a stage's k-steps are dealt over the waves");
    constexpr int BW = SEG / NW;                     // bytes of a row's stage segment per wave: 256 / 128
    constexpr int LPR = BW / 16;                     // lanes (16-byte slots) per row of a request: 16 / 8
    constexpr int RPP = 64 / LPR;                    // rows per request: 4 / 8
    constexpr int PP = RT / RPP;                     // requests per wave and stage: 4 / 2
    constexpr int WREG = RT * BW;                    // a wave's share of a ring slot: 4 KiB / 2 KiB
    constexpr int RED_BYTES = mfw_red_bytes(NW);
    constexpr int KMINE = KCH * KPW;                 // k-steps (= fragments per query block) of one wave
    static_assert(NS - 1 <= KCH && (NS - 2) * PP + 2 <= 33, "requests reach into the next tile at most");
    static_assert(NQ <= 4 || (NW == 4 && NQ % 4 == 0), "column block nt is collected by wave nt % NW");
    constexpr int OWN = NW == 4 ? 4 : NQ;   // the waves that collect column blocks (block nt: wave nt % OWN); eight waves: NQ <= 4
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m16 = lane & 15;
    const int kq = lane >> 4;
    const int qtile = blockIdx.y;

    // this wave's fragments: k-steps wave * KPW .. wave * KPW + KPW - 1 of every stage
    bf16x8_t qf[NQ][KMINE];
    int qidx[NQ];
    float nq2[NQ], tau[NQ], qnorm[NQ];
#pragma unroll
    for (int nt = 0; nt < NQ; nt++) {
        const uint4 *src = P.qfrag + ((size_t)(qtile * NQ + nt) * KSTEPS) * 64 + lane;
#pragma unroll
        for (int i = 0; i < KMINE; i++) {
            const int s = (i / KPW) * KSUB + wave * KPW + (i % KPW);   // stage i / KPW, its k-step wave * KPW + i % KPW
            uint4 v = src[(size_t)s * 64];
            qf[nt][i] = __builtin_bit_cast(bf16x8_t, v);
        }
        qidx[nt] = (qtile * NQ + nt) * MFW_QTILE + m16;
        nq2[nt] = P.qn2[qidx[nt]];
        qnorm[nt] = U8C ? P.qmeta[(size_t)qidx[nt] * 8] : 0.f;
        tau[nt] = MODE == MF_FILTER ? P.tau[qidx[nt]] : 0.f;
    }
#pragma unroll
    for (int nt = 0; nt < NQ; nt++) {
#pragma unroll
        for (int i = 0; i < KMINE; i++) {
            // more than ~220 registers of fragments (four column blocks at KSTEPS 96, three at 128: 384): the upper half -- by fragment,
            // not by block -- is parked in AGPRs, which the MFMA reads directly (one wave per SIMD owns all 512 registers of a lane, one
            // of two 256); left to hipcc they are spilled
            if (NQ * KMINE * 4 > (NW == 8 ? 100 : 224) && 2 * (nt * KMINE + i) >= NQ * KMINE) asm volatile("" : "+a"(qf[nt][i]));
            else asm volatile("" : "+v"(qf[nt][i]));
        }
        asm volatile("" : "+v"(nq2[nt]), "+v"(tau[nt]), "+v"(qnorm[nt]));
    }

    // request t of this wave: rows RPP t .. RPP t + RPP - 1 of the tile, BW bytes each (LPR lanes x 16 B); lane l -> row RPP t + l / LPR,
    // LDS slot l % LPR of the row's block, which takes the 16 bytes at slot ^ swz(row) of the block in memory (the fragment reads of 16
    // rows BW bytes apart then spread over the banks: swz = row % 16 for 16 slots; for 8 slots row % 8 ^ row / 8 -- rows r and r + 8 lie
    // 1 KiB apart, on the same banks)
    // The request's address is a wave-uniform base in SGPRs -- the tile's first row, the stage's column -- plus this constant 32-bit offset
    // of the lane (`global_load_lds_dwordx4 v, s[..]`: no 64-bit VGPR address per request, none to rebuild per tile).  Rows past the
    // table's end are not clamped: a tile never leaves its slab, slabs are allocated whole, the epilogue masks such rows.
    auto swz = [](uint32_t row) -> uint32_t { return NW == 4 ? (row & 15u) : ((row & 7u) ^ (row >> 3)); };
    uint32_t voff[PP];
#pragma unroll
    for (int t = 0; t < PP; t++) {
        const uint32_t row = (uint32_t)(RPP * t) + (uint32_t)lane / (uint32_t)LPR, slot = (uint32_t)lane % (uint32_t)LPR;
        voff[t] = row * P.row_stride + (uint32_t)BW * (uint32_t)wave + ((slot ^ swz(row)) * 16u);
    }
    const uint32_t lds_stage_wave_off = (uint32_t)(wave * WREG);   // this wave's share of a ring slot: 16 rows x BW bytes
    char *norm_lds = lds + NS * MF_STAGE_BYTES;
    const bool norm_loader = wave == 0;
    uint32_t *eq_n = reinterpret_cast<uint32_t *>(lds + NS * MF_STAGE_BYTES + NORM_BYTES);
    uint4 *eq = reinterpret_cast<uint4 *>(lds + NS * MF_STAGE_BYTES + NORM_BYTES + 16);
    const uint32_t eq_n_off = mf_lds_offset(eq_n), eq_off = mf_lds_offset(eq);
    const uint32_t red_off = mf_lds_offset(lds + NS * MF_STAGE_BYTES + NORM_BYTES + MF_EQ_BYTES);   // two tiles' partial sums
    if (MODE == MF_FILTER && tid == 0) *eq_n = 0;

    const uint32_t step = gridDim.x;
    auto tile_row0 = [&](uint32_t t) -> uint32_t {
        return (P.tile_first + (t >> P.tile_run_shift) * (P.tile_step << P.tile_run_shift) + (t & ((1u << P.tile_run_shift) - 1u))) * RT;
    };
    uint64_t rb_cur, rb_nxt;   // (scalar) address of the tile's first row
    const float *np_cur, *np_nxt;
    uint32_t cur_slab = 0xFFFFFFFFu;
    uint64_t cur_sbase = 0, cur_nbase = 0;
    auto make_ptrs = [&](uint32_t t, uint64_t &rb, const float *&np) {
        uint32_t tt = t < P.n_tiles ? t : P.n_tiles - 1;
        const uint32_t r0 = tile_row0(tt);
        const uint32_t sidx = __builtin_amdgcn_readfirstlane(r0 >> P.slab_shift);
        if (sidx != cur_slab) {   // scalar loads: a vector load here would put a vmcnt(0) into every tile
            cur_slab = sidx;
            const char *const *sp = P.slabs + sidx;
            const float *const *npp = P.norm_slabs + sidx;
            asm volatile("s_load_dwordx2 %0, %2, 0x0\n\ts_load_dwordx2 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&s"(cur_sbase), "=&s"(cur_nbase)
                         : "s"(sp), "s"(npp)
                         : "memory");
        }
        const float *nbase = reinterpret_cast<const float *>(cur_nbase);
        const uint32_t rs = (uint32_t)__builtin_amdgcn_readfirstlane((int)(r0 & P.slab_mask));
        rb = cur_sbase + (uint64_t)rs * (uint64_t)P.row_stride;
        uint32_t nrow = r0 + lane;
        if (nrow >= P.n_rows) nrow = P.n_rows - 1;
        np = nbase + (size_t)(nrow & P.slab_mask) * (U8C ? 4 : 1);
    };
    const uint32_t lds_ring_off = mf_lds_offset(lds) + lds_stage_wave_off;
    auto issue = [&](uint64_t rb, const float *np, int kc, uint32_t slot, bool with_norm, uint32_t norm_buf) {
        const uint32_t base = lds_ring_off + slot * MF_STAGE_BYTES;
        const uint64_t sb = rb + (uint64_t)(kc * SEG);
#pragma unroll
        for (int i = 0; i < PP; i++) {   // LDS address = M0 + the lane's 16 bytes
            if constexpr (AUX == 2)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff[i]), "s"(sb), "s"(base + i * 1024) : "memory", "m0");
            else
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff[i]), "s"(sb), "s"(base + i * 1024) : "memory", "m0");
        }
        if (with_norm && norm_loader) {
            if constexpr (U8C) glds16<0>(np, norm_buf * NORM_PAR, norm_lds);   // 64 records of 16 bytes (the tile's 16 in front)
            else glds4(np, norm_buf * NORM_PAR, norm_lds);
        }
    };

    uint32_t tile = blockIdx.x;
    make_ptrs(tile, rb_cur, np_cur);
    make_ptrs(tile + step, rb_nxt, np_nxt);
    uint32_t slot_c = 0, nbuf = 0, rpar = 0;   // ring slot of the current stage; aux buffer (of three) and partial-sum buffer (of two) of the current tile
#pragma unroll
    for (int u = 0; u < NS - 1; u++) issue(rb_cur, np_cur, u, u, u == 0, 0);

    const uint32_t pm_off = mf_lds_offset(lds) + (uint32_t)mfw_lds_bytes(false, NQ, NS, EK) + (uint32_t)m16 * 4u;
    uint32_t pm_n = 0, pm_tile0 = 0;
    auto flush_probe_minima = [&]() {   // (a wave that collects column blocks: the buffered minima of its blocks)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int nt = 0; nt < NQ; nt++) {
            if (wave != nt % OWN) continue;
            for (uint32_t it = (uint32_t)kq; it < pm_n; it += 4) {
                float v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(pm_off + (it * NQ + nt) * 64u) : "memory");
                P.tilemin[(size_t)qidx[nt] * P.tilemin_stride + pm_tile0 + it * step] = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pm_n = 0;
    };
    // One k-step's A operand from this wave's quarter of a ring slot, and its MFMAs.
    //   8-bit rows: a k-step is 64 elements = 64 bytes of the row, read like a 16-bit k-step; uint8 rows ride the signed MFMA re-centred
    //               by 128 (the query fragments were re-centred on the host: DESIGN.md 5.5)
    //   fp32 rows:  a k-step is 32 elements = 128 bytes, lane (m16, kq) converts its 8
    //   16-bit rows: a k-step is 64 bytes of the row, lane (m16, kq) reads its 16 (slot (4 jj + kq) ^ m16 of the row's block)
    // The operand reads are inline asm with COUNTED waits: behind the asm waits of this loop hipcc's own scoreboard starts from "unknown"
    // and put a full lgkmcnt(0) in front of the first MFMA after every fresh pair of reads -- the matrix pipe idle for an LDS round trip
    // twice per unit.  LDS operations return in order, so "all but the newest n" is exact.
    using a_t = typename std::conditional<EK == 0, bf16x8_t, mf_u32x4>::type;
    constexpr int RD = EK == 0 ? 2 : 1;   // LDS reads per operand
    struct araw {
        mf_u32x4 lo, hi;   // (hi: fp32 rows only)
    };
    const uint32_t rd_base = mf_lds_offset(lds) + lds_stage_wave_off + (uint32_t)m16 * (uint32_t)BW;
    const uint32_t rd_sw = swz((uint32_t)m16);
    auto read_a = [&](uint32_t slot, int jj) -> araw {   // issues the reads; wait_a() before finish_a()
        araw r;
        const uint32_t rowp = rd_base + slot * MF_STAGE_BYTES;
        if constexpr (EK == 0) {
            const uint32_t p0 = (uint32_t)(8 * jj + 2 * kq) ^ rd_sw, p1 = (uint32_t)(8 * jj + 2 * kq + 1) ^ rd_sw;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(r.lo), "=&v"(r.hi) : "v"(rowp + p0 * 16u), "v"(rowp + p1 * 16u) : "memory");
        } else {
            const uint32_t p = (uint32_t)(4 * jj + kq) ^ rd_sw;
            asm volatile("ds_read_b128 %0, %1" : "=v"(r.lo) : "v"(rowp + p * 16u) : "memory");
            r.hi = r.lo;
        }
        return r;
    };
    auto finish_a = [&](const araw &r) -> a_t {
        if constexpr (EK == 0) {
            f32x8_t x = __builtin_shufflevector(__builtin_bit_cast(f32x4_t, r.lo), __builtin_bit_cast(f32x4_t, r.hi), 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_convertvector(x, bf16x8_t);
        } else if constexpr (EK >= 4) {
            return r.lo ^ mf_u32x4{0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};
        } else {
            return r.lo;
        }
    };
    // PF operands of the NEXT stage are read while this stage's MFMAs run (nothing but the wave's own vmcnt says when a stage is
    // there), so a unit starts multiplying at once instead of behind an LDS round trip
    constexpr int PF = KPW >= 4 ? 2 : 1;
    constexpr int NPF = PF * RD;             // LDS reads behind the prefetched operands
    constexpr int NREST = (KPW - PF) * RD;   // ... behind the stage's other operands
    araw apf[PF];
    mfw_wait_vmcnt((NS - 2) * PP);   // stage 0 (and, wave 0, its aux values): NS - 2 younger stages in flight
#pragma unroll
    for (int j = 0; j < PF; j++) apf[j] = read_a(0, j);
    for (; tile < P.n_tiles; tile += step) {
        acc_v acc[NQ];
#pragma unroll
        for (int nt = 0; nt < NQ; nt++) acc[nt] = acc_v{0, 0, 0, 0};
        auto mma = [&](a_t a, int fi) {
#pragma unroll
            for (int nt = 0; nt < NQ; nt++) {
                if constexpr (INT8)
                    acc[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4w_t, a), __builtin_bit_cast(i32x4w_t, qf[nt][fi]), acc[nt], 0, 0, 0);
                else if constexpr (EK == 2)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, qf[nt][fi]), acc[nt], 0, 0, 0);
                else
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), qf[nt][fi], acc[nt], 0, 0, 0);
            }
        };
#pragma unroll
        for (int c = 0; c < KCH; c++) {
            {   // the slot this wave finished with the previous stage takes the stage NS - 1 ahead.  Its reads of that slot are complete: the
                // only LDS operations issued since they were waited for are the NPF reads that fetched THIS stage's first operands
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NPF) : "memory");
                constexpr int D = NS - 1;
                uint32_t slot_p = slot_c + D;
                if (slot_p >= NS) slot_p -= NS;
                const int cc = c + D;
                const uint32_t nb1 = nbuf + 1 >= 3 ? 0 : nbuf + 1;
                if (cc < KCH) issue(rb_cur, np_cur, cc, slot_p, false, 0);
                else issue(rb_nxt, np_nxt, cc - KCH, slot_p, cc == KCH, nb1);
            }
            // this wave's k-steps of the stage, from its own quarter of the slot: the first PF operands were requested a unit ago
            araw ar[KPW];
#pragma unroll
            for (int jj = 0; jj < KPW; jj++) ar[jj] = jj < PF ? apf[jj] : read_a(slot_c, jj);
#pragma unroll
            for (int jj = 0; jj < PF; jj++)   // all but the NREST reads just issued have returned
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ar[jj].lo), "+v"(ar[jj].hi) : "n"(NREST) : "memory");
            mma(finish_a(ar[0]), c * KPW);
            {   // this wave's part of stage c + 1 landed?  Its NS - 2 younger stages (PP loads each, + the aux load of a stage that opens a
                // tile) may be in flight.  Then its first operands, under the MFMAs just issued
                int n_norm = 0;
#pragma unroll
                for (int j = 2; j < NS; j++) n_norm += ((c + j) % KCH == 0) ? 1 : 0;
                if (norm_loader) mfw_wait_vmcnt((NS - 2) * PP + n_norm);
                else mfw_wait_vmcnt((NS - 2) * PP);
                const uint32_t slot_n = slot_c + 1 == NS ? 0 : slot_c + 1;
#pragma unroll
                for (int j = 0; j < PF; j++) apf[j] = read_a(slot_n, j);
            }
#pragma unroll
            for (int jj = 1; jj < PF; jj++) mma(finish_a(ar[jj]), c * KPW + jj);
#pragma unroll
            for (int jj = PF; jj < KPW; jj++) {   // all but the NPF reads of the next stage have returned
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ar[jj].lo), "+v"(ar[jj].hi) : "n"(NPF) : "memory");
                mma(finish_a(ar[jj]), c * KPW + jj);
            }
            slot_c = slot_c + 1 == NS ? 0 : slot_c + 1;
        }
        // ---- the four waves' partial dot products of (row kq*4 + i, query m16) meet in LDS: wave nt collects column block nt ----
        const uint32_t red_tile = red_off + rpar * (uint32_t)(NQ * RED_BYTES);
#pragma unroll
        for (int nt = 0; nt < NQ; nt++) {
            if (wave != nt % OWN) {
                const int k3 = wave < nt % OWN ? wave : wave - 1;   // which of the NW - 1 partials of block nt this wave writes
                const mf_u32x4 v = __builtin_bit_cast(mf_u32x4, acc[nt]);
                asm volatile("ds_write_b128 %0, %1" ::"v"(red_tile + (uint32_t)(nt * RED_BYTES + k3 * 1024 + lane * 16)), "v"(v) : "memory");
            }
        }
        // The one barrier of a row tile.  Behind it every wave has left the previous tile's epilogue, so: the queue length is final and
        // uniform (flush decision), the other partial-sum buffer may be written again by the next tile, and the aux buffer this tile's
        // successor will have refilled two tiles on is no longer read (three aux buffers: wave 0 requests tile t + 1's values during
        // tile t, i.e. possibly while other waves are still in the epilogue of tile t - 1).
        mf_ring_barrier();
        if (MODE == MF_FILTER) {
            if (*eq_n >= MF_EQ_CAP / 2) mf_flush_queue<64 * NW>(eq_n, eq, P.counts, P.cand, P.cap);
        }
        const uint32_t r0 = tile_row0(tile);
        bool emitted = false;
#pragma unroll
        for (int nt = 0; nt < NQ; nt++) {
            if (wave != nt % OWN) continue;
            const float *nrm = reinterpret_cast<const float *>(norm_lds + nbuf * NORM_PAR);
            mf_u32x4 nbits, sxbits = {0u, 0u, 0u, 0u};
            if constexpr (U8C) {   // rows kq * 4 + i: the record's first two words
                u32x2_t r2[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    asm volatile("ds_read_b64 %0, %1" : "=v"(r2[i]) : "v"(mf_lds_offset(nrm) + (uint32_t)((kq * 4 + i) * 16)) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r2[0]), "+v"(r2[1]), "+v"(r2[2]), "+v"(r2[3]));
#pragma unroll
                for (int i = 0; i < 4; i++) nbits[i] = r2[i][0], sxbits[i] = r2[i][1];
            } else {
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nbits) : "v"(mf_lds_offset(nrm) + (uint32_t)(kq * 16)) : "memory");
            }
            const f32x4_t n4 = __builtin_bit_cast(f32x4_t, nbits);
            acc_v a4 = acc[nt];
#pragma unroll
            for (int g = 0; g + 3 <= NW - 1; g += 3) {   // the other waves' partials, three at a time, in wave order
                mf_u32x4 p1, p2, p3;
                asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:1024\n\tds_read_b128 %2, %3 offset:2048\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(p1), "=&v"(p2), "=&v"(p3)
                             : "v"(red_tile + (uint32_t)(nt * RED_BYTES + g * 1024 + lane * 16))
                             : "memory");
                a4 += __builtin_bit_cast(acc_v, p1);
                a4 += __builtin_bit_cast(acc_v, p2);
                a4 += __builtin_bit_cast(acc_v, p3);
            }
            if constexpr ((NW - 1) % 3 == 1) {   // (eight waves: the seventh)
                mf_u32x4 p1;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p1) : "v"(red_tile + (uint32_t)(nt * RED_BYTES + (NW - 2) * 1024 + lane * 16)) : "memory");
                a4 += __builtin_bit_cast(acc_v, p1);
            }
            static_assert((NW - 1) % 3 != 2, "partials are read in threes, plus one");
            float tmin = INFINITY;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t row = r0 + kq * 4 + i;
                float low, up;
                bool pass;
                if constexpr (INT8) {
                    // exact integer dot: the reference's own score (mfma_lowp_kernels.hpp has the same four epilogues); aux =
                    // sum x^2 (L2; of the re-centred bytes for uint8), sum x' (uint8 IP) or the stored float norm (Cosine)
                    int dot = (int)a4[i];
                    const uint32_t av = nbits[i];
                    uint32_t qa = __float_as_uint(nq2[nt]);
                    if constexpr (U8C) {   // sum x q = sum x'q' + 128 sum x' + (128 sum q' + 128^2 dim); the score from the two stored norms
                        dot += 128 * (int)sxbits[i] + (int)qa;
                        qa = __float_as_uint(qnorm[nt]);
                    }
                    float sc;
                    if (P.iepi == 2) sc = (float)((int)av + (int)qa - 2 * dot);
                    else if (P.iepi == 3) sc = (float)(1 - dot);
                    else if (P.iepi == 5) sc = (float)(1 - (dot + 128 * (int)av + (int)qa));
                    else sc = __fsub_rn(1.0f, __fdiv_rn((float)dot, __fmul_rn(__uint_as_float(av), __uint_as_float(qa))));
                    low = up = sc;
                    pass = sc <= tau[nt];
                } else {
                    const float ssum = n4[i] + nq2[nt];
                    const float dot = (float)a4[i];
                    const float a = P.is_l2 ? (ssum - 2.0f * dot) : (1.0f - dot);
                    const float E = P.cE * ssum + P.absE;
                    low = a - E;
                    up = a + E;
                    pass = !(low > tau[nt]);
                }
                if (MODE == MF_PROBE) {
                    if (row < P.n_rows && up < tmin) tmin = up;
                } else {
                    if (row < P.n_rows && pass) {
                        const uint32_t pos = mf_queue_reserve(eq_n_off);
                        if (pos < MF_EQ_CAP) {
                            mf_queue_write(eq_off + pos * 16, row, (uint32_t)qidx[nt], __float_as_uint(low));
                        } else {
                            uint32_t s = atomicAdd(&P.counts[qidx[nt]], 1u);
                            if (s < P.cap) P.cand[(size_t)qidx[nt] * P.cap + s] = make_uint2(row, __float_as_uint(low));
                            emitted = true;
                        }
                    }
                }
            }
            if (MODE == MF_PROBE) {
                tmin = fminf(tmin, __shfl_xor(tmin, 16));
                tmin = fminf(tmin, __shfl_xor(tmin, 32));
                if (kq == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(pm_off + (pm_n * NQ + nt) * 64u), "v"(tmin) : "memory");
            }
        }
        if (MODE == MF_PROBE && wave < (NQ < OWN ? NQ : OWN)) {   // (once per tile, whatever the number of blocks the wave collects)
            if (pm_n == 0) pm_tile0 = tile;
            if (++pm_n == (uint32_t)MF_PM_TILES) flush_probe_minima();
        }
        if (MODE == MF_FILTER) {
            if (__any(emitted)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        rb_cur = rb_nxt;
        np_cur = np_nxt;
        make_ptrs(tile + 2 * step, rb_nxt, np_nxt);
        nbuf = nbuf + 1 >= 3 ? 0 : nbuf + 1;
        rpar ^= 1u;
    }
    if (MODE == MF_PROBE && pm_n) flush_probe_minima();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (MODE == MF_FILTER) {
        __builtin_amdgcn_s_barrier();
        mf_flush_queue<64 * NW>(eq_n, eq, P.counts, P.cand, P.cap);
    }
}

}  // namespace vsg
