// LDS-DMA k loop for the 256 x 256-tile weight-gradient GEMMs (C = A^T B, both operands stored [k][x]) on operands that
// already carry the bf16 hi / lo split (XPS_FMT_SPLIT4, include/xps.h: per aligned group of four fp32 elements the same
// 16 bytes hold hi[0..3] | lo[0..3]).  The register-staged loop of xps_gemm_big.h moves every operand element
// global -> VGPR -> (split) -> ds_write -> ds_read_tr -> MFMA; here a k row of a tile (256 x elements = 1 KiB) is ONE
// LDS-DMA piece (global_load_lds_dwordx4: 64 lanes x 16 B, source = the row as it lies in memory, full 128-byte lines) and the
// loop is DMA + transposing LDS reads + MFMA: no staging registers, no split arithmetic, no ds_write.
//
// Why the split4 format can be read in place: ds_read_b64_tr_b16 takes, per lane, the address of four consecutive x elements
// of one k row (8 bytes) -- exactly the hi half (bytes 0-7) or the lo half (bytes 8-15) of a split4 group.  (The [x][k]
// operands of the NT / NN forms need 8 consecutive k per fragment register pair and cannot be read in place: they keep the
// register-staged loop.)
//
// LDS image of one operand and k-tile: 16 rows of 1 KiB.  A transposed read of a 32-lane half touches 4 rows x 8 groups and
// takes the SAME 8 of each group's 16 bytes: conflict-free only when the four rows sit at byte offsets {0, 8, 128, 136} mod
// 256 -- which a DMA piece's wave-uniform destination (M0) can give them: row k at  1024 k + 144 (k / 4) + {0, 8, 128, 136}[k % 4].
//
// Ring of DMA_D = 4 stages (k-tiles): at iteration kt the pieces of k-tiles kt + 1, kt + 2 are in flight and kt + 3 is issued
// right behind the barrier that freed its stage (the one read at kt - 1): 96 KiB in flight per CU, one barrier per k-tile,
// counted s_waitcnt vmcnt (never 0 in the loop).  Same MFMA order per accumulator as every other tile kernel (per k-tile
// lo*hi, hi*lo, hi*hi), so the products have the same bits.
#pragma once
#include <type_traits>
#include "xps_gemm_big.h"

namespace xps_big {

#ifndef XPS_DMA_STAGES
#define XPS_DMA_STAGES 4
#endif
constexpr int DMA_D = XPS_DMA_STAGES;
constexpr int DMA_ROW = 1024;
constexpr int DMA_QPAD = 144;
constexpr int DMA_IMG = 16 * DMA_ROW + 4 * DMA_QPAD;          // 16960 B: one operand, one k-tile
constexpr int DMA_STAGE = 2 * DMA_IMG;                        // A then B
constexpr int DMA_LDS = DMA_D * DMA_STAGE;                    // 135680 B
#ifndef XPS_DMA_SHIFT8
#define XPS_DMA_SHIFT8 1
#endif
__host__ __device__ constexpr int dma_rowoff(int k) {
    return k * DMA_ROW + (k >> 2) * DMA_QPAD + ((k & 1) ? (XPS_DMA_SHIFT8 ? 8 : 0) : 0) + ((k & 2) ? 128 : 0);
}
constexpr int DMA_ROW4 = 4 * DMA_ROW + DMA_QPAD;              // rowoff(k + 4) - rowoff(k)

#ifdef XPS_DMA_STAMP
// diagnostic build only (tools/proto/tn_dma.hip): per-wave cycle sums of the loop segments, in a buffer no kernel reads
__device__ unsigned long long g_dma_stamp[4096 * 4];
#endif

__device__ inline unsigned dma_lds_base(const unsigned char* smem) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) unsigned char*)smem;
}

// one 1-KiB piece: 64 lanes x 16 B from (sbase + voff) to LDS [ldst, ldst + 1024); sbase / ldst wave-uniform
__device__ inline void dma_piece(const unsigned char* sbase, unsigned voff, unsigned ldst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(ldst)
                 : "memory");
}

#ifndef XPS_DMA_SPREAD
#define XPS_DMA_SPREAD 1
#endif
#ifndef XPS_DMA_STAGGER
#define XPS_DMA_STAGGER 0
#endif
// XPS_DMA_PAIR=1: two k-tiles per barrier (below).  Measured level with one barrier per k-tile (tools/proto/tn_dma.hip, dW_ih of
// configs[3], two interleaved runs each: 318.7 / 321.5 us against 319.5 / 321.5 us, same bits) -- like the staggered and the
// merged-barrier forms before it: with 65 % of the cycles on the matrix pipe at the 1.77 GHz the power limit leaves, removing
// idle cycles does not buy time.  Compiled out.
#ifndef XPS_DMA_PAIR
#define XPS_DMA_PAIR 0
#endif

// acc += A[kbeg .. kbeg + 16 nkt)^T B[...] for the 256 x 256 tile at (m0, n0); A, B: split4 operands, [k][x], leading
// dimensions lda / ldb (elements).  Ends with every DMA piece landed and a barrier (LDS is free).
// CS: also column sums of the A tile (the bias gradient), ON THE MATRIX PIPE: cacc += A_g^T 1 for the 32-column group
// g = 4 (wave / 4) + wave % 4 of the tile's 256 (two MFMAs per k-tile, lo then hi, against an all-ones operand), in the waves whose
// group lies in [cs_lo, cs_hi): the n-tile blocks that share an A tile SHARE its eight groups (tiles_n = 4: two groups per
// block), so that no block carries all of the extra work.  Every column of the 32 x 32 result holds the same 32 sums;
// dma_colsum_store writes column 0 of the groups this block owns.  Measured on the way (tools/proto/tn_dma.hip, dW_ih of
// configs[3], every block summing all eight groups): from the LDS image on the vector ALU (16 conversions + 12 adds per thread
// and k-tile) + 27 % however the instructions were placed; on the matrix pipe + 21 % for + 8 % MFMAs -- the stamped builds show
// + 10 % cycles and an 11 % lower clock (1.56 -> 1.38 GHz): the loop is power-limited, extra work costs more than its cycles.
//
// Schedule of a k-tile (pinned with scheduling barriers).  Stamps of the first form -- all four pieces right behind the
// barrier, then reads + MFMAs -- read per k-tile and wave: vmcnt wait 45 cycles (the ring is deep enough), barrier 464, DMA
// issue 357, reads + MFMA 1244 (floor 768 per wave, 1536 per SIMD): both waves of a SIMD spent the same ~360 cycles issuing
// pieces while its matrix pipe idled.  Now: fragment requests of B and row group 0, two pieces in the shadow of that LDS
// latency, then per row group the requests of the next group, six MFMAs and (groups 0, 1) one more piece -- a wave that is
// issuing a piece leaves the pipe to its partner.
template <bool CS>
__device__ inline void tn_dma_pipeline(f32x16 (&acc)[4][2], f32x16& cacc, const float* __restrict__ A, long long lda,
                                       const float* __restrict__ B, long long ldb, int m0, int n0, int kbeg, int nkt,
                                       unsigned char* smem, int cs_lo = 0, int cs_hi = 8) {
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    const unsigned lds0 = dma_lds_base(smem);
    if (nkt <= 0) return;
    // ---- DMA roles: wave w moves rows 2 w, 2 w + 1 of both operands
    const unsigned va0 = (unsigned)((long long)(2 * wave) * lda * 4 + lane * 16), va1 = va0 + (unsigned)(lda * 4);
    const unsigned vb0 = (unsigned)((long long)(2 * wave) * ldb * 4 + lane * 16), vb1 = vb0 + (unsigned)(ldb * 4);
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(A + (long long)kbeg * lda + m0);
    const unsigned char* bbase = reinterpret_cast<const unsigned char*>(B + (long long)kbeg * ldb + n0);
    const long long stepa = 16 * lda * 4, stepb = 16 * ldb * 4;
    const unsigned ra0 = (unsigned)dma_rowoff(2 * wave), ra1 = (unsigned)dma_rowoff(2 * wave + 1);
    // piece pc (0, 1: rows of A; 2, 3: rows of B) of k-tile kt (beyond the range: the last k-tile again, into a stage nobody reads)
    auto piece = [&](int kt, int pc) {
        const int ks = kt < nkt ? kt : nkt - 1;
        const unsigned st = lds0 + (unsigned)(kt % DMA_D) * DMA_STAGE;
        if (pc == 0) dma_piece(abase + ks * stepa, va0, st + ra0);
        else if (pc == 1) dma_piece(abase + ks * stepa, va1, st + ra1);
        else if (pc == 2) dma_piece(bbase + ks * stepb, vb0, st + DMA_IMG + ra0);
        else dma_piece(bbase + ks * stepb, vb1, st + DMA_IMG + ra1);
    };
    // ---- fragment addresses (32 x 32 x 16 operand: lane -> x = x0 + (lane & 31), k = 8 (lane >> 5) + 0..7; two transposed
    //      reads of 4 k rows x 16 x per 16-lane group: lane 4 q + p of group g supplies row 8 (g / 2) + q (+ 4), x group p)
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int fro = dma_rowoff(8 * (g >> 1) + q) + ((g & 1) * 4 + pp) * 16;
    const int fa = fro + (wm / 4) * 16, fb = DMA_IMG + fro + (wn / 4) * 16;
    const int csel = wave & 3, cgrp = (wave >> 2) * 4 + csel;
    const bool do_cs = CS && cgrp >= cs_lo && cgrp < cs_hi;           // (wave-uniform)
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
#pragma unroll
    for (int d = 0; d < (XPS_DMA_PAIR && DMA_D == 4 && !XPS_DMA_STAGGER ? 2 : DMA_D - 1); ++d)
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) piece(d, pc);
#ifdef XPS_DMA_STAMP
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, s_wait = 0, s_bar = 0, s_dma = 0, s_mma = 0;
#define DSTAMP(var) __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); __builtin_amdgcn_sched_barrier(0);
#else
#define DSTAMP(var)
#endif
    auto frag = [&](const unsigned char* p, bf16x8& fh, bf16x8& fl) {
        const bf16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
        const bf16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + DMA_ROW4));
        const bf16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 8));
        const bf16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + DMA_ROW4 + 8));
        fh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        fl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // the six MFMAs of row group i (+ the two of the column sums where this wave owns the group)
    auto mma_group = [&](auto I, const bf16x8& ah, const bf16x8& al, const bf16x8 (&bh)[2], const bf16x8 (&bl)[2]) {
        constexpr int i = decltype(I)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
        }
        if constexpr (CS) {
            if (i == csel && do_cs) {
                cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ones, cacc, 0, 0, 0);
                cacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ones, cacc, 0, 0, 0);
            }
        }
    };
    auto wait_tile = [&]() {
        if constexpr (DMA_D == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // this wave's pieces of k-tile kt have landed
        else if constexpr (DMA_D == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (XPS_DMA_STAGGER && wave >= 4) {
        // Waves 4-7 (the SIMD partners of waves 0-3) run HALF A K-TILE BEHIND: the MFMAs of row groups 2, 3 of a k-tile are issued
        // after the NEXT barrier, from fragments read before it (every LDS read of a stage still precedes the barrier that frees
        // it).  Right behind a barrier a SIMD then has one wave with twelve MFMAs ready in registers while its partner waits for
        // its first fragments, instead of two waves waiting for LDS with an idle matrix pipe (stamps: ~500 of ~2050 cycles per
        // k-tile; MI355X_MICROARCH.md, two waves per SIMD, item 9).  Same MFMA order per accumulator: same bits.
        struct Held { bf16x8 bh[2], bl[2], a2h, a2l, a3h, a3l; };
        Held H0, H1;
        auto late = [&](int kt, Held& cur, const Held& prev) {
            wait_tile();
            __builtin_amdgcn_s_barrier();
            const unsigned char* st = smem + (kt % DMA_D) * DMA_STAGE;
            bf16x8 a0h, a0l, a1h, a1l;
#pragma unroll
            for (int j = 0; j < 2; ++j) frag(st + fb + j * 128, cur.bh[j], cur.bl[j]);
            frag(st + fa, a0h, a0l);
            __builtin_amdgcn_sched_barrier(0);
            piece(kt + DMA_D - 1, 0); piece(kt + DMA_D - 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (kt > 0) mma_group(std::integral_constant<int, 2>{}, prev.a2h, prev.a2l, prev.bh, prev.bl);
            frag(st + fa + 128, a1h, a1l);
            __builtin_amdgcn_sched_barrier(0);
            if (kt > 0) mma_group(std::integral_constant<int, 3>{}, prev.a3h, prev.a3l, prev.bh, prev.bl);
            __builtin_amdgcn_sched_barrier(0);
            piece(kt + DMA_D - 1, 2);
            __builtin_amdgcn_sched_barrier(0);
            frag(st + fa + 256, cur.a2h, cur.a2l);
            mma_group(std::integral_constant<int, 0>{}, a0h, a0l, cur.bh, cur.bl);
            __builtin_amdgcn_sched_barrier(0);
            piece(kt + DMA_D - 1, 3);
            __builtin_amdgcn_sched_barrier(0);
            frag(st + fa + 384, cur.a3h, cur.a3l);
            mma_group(std::integral_constant<int, 1>{}, a1h, a1l, cur.bh, cur.bl);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the held fragments are in registers before the stage is released
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int kt = 0; kt < nkt; kt += 2) {
            late(kt, H0, H1);
            if (kt + 1 < nkt) late(kt + 1, H1, H0);
        }
        if ((nkt - 1) & 1) {
            mma_group(std::integral_constant<int, 2>{}, H1.a2h, H1.a2l, H1.bh, H1.bl);
            mma_group(std::integral_constant<int, 3>{}, H1.a3h, H1.a3l, H1.bh, H1.bl);
        } else {
            mma_group(std::integral_constant<int, 2>{}, H0.a2h, H0.a2l, H0.bh, H0.bl);
            mma_group(std::integral_constant<int, 3>{}, H0.a3h, H0.a3l, H0.bh, H0.bl);
        }
    } else if (XPS_DMA_PAIR && DMA_D == 4) {
        // TWO k-tiles per barrier: the ring is two half-rings of two stages; behind the barrier that ends pair P - 1 the eight
        // pieces of pair P + 1 go out (into the stages pair P - 1 was read from) while pair P is multiplied -- 64 KiB in flight per
        // CU for two k-tile times.  Halves the barriers (stamps of the one-k-tile form: 464 of ~2100 cycles per k-tile and wave at
        // the barrier, mostly skew between the eight waves) and the fragment-latency bubbles behind them (the second k-tile's
        // fragments are requested under the first one's last MFMAs).  Same MFMA order per accumulator: same bits.
        for (int kt = 0; kt < nkt; kt += 2) {
            DSTAMP(t0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of k-tiles kt, kt + 1 have landed
            DSTAMP(t1)
            __builtin_amdgcn_s_barrier();                          // ... everybody's; and everybody is done reading kt - 2, kt - 1
            DSTAMP(t2)
            DSTAMP(t3)
            const bool two = kt + 1 < nkt, more = kt + 2 < nkt, more2 = kt + 3 < nkt;      // (uniform)
            const unsigned char* st0 = smem + (kt % DMA_D) * DMA_STAGE;
            const unsigned char* st1 = smem + ((kt + 1) % DMA_D) * DMA_STAGE;
            bf16x8 bh[2], bl[2], ah[2], al[2], ch[2], cl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) frag(st0 + fb + j * 128, bh[j], bl[j]);
            frag(st0 + fa, ah[0], al[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (more) { piece(kt + 2, 0); piece(kt + 2, 1); }
            __builtin_amdgcn_sched_barrier(0);
            // k-tile kt: row group i's six MFMAs under the requests of group i + 1 (group 3: of k-tile kt + 1's first fragments)
            frag(st0 + fa + 128, ah[1], al[1]);
            mma_group(std::integral_constant<int, 0>{}, ah[0], al[0], bh, bl);
            __builtin_amdgcn_sched_barrier(0);
            if (more) { piece(kt + 2, 2); piece(kt + 2, 3); }
            __builtin_amdgcn_sched_barrier(0);
            frag(st0 + fa + 256, ah[0], al[0]);
            mma_group(std::integral_constant<int, 1>{}, ah[1], al[1], bh, bl);
            __builtin_amdgcn_sched_barrier(0);
            if (more2) { piece(kt + 3, 0); piece(kt + 3, 1); }
            __builtin_amdgcn_sched_barrier(0);
            frag(st0 + fa + 384, ah[1], al[1]);
            mma_group(std::integral_constant<int, 2>{}, ah[0], al[0], bh, bl);
            __builtin_amdgcn_sched_barrier(0);
            if (more2) { piece(kt + 3, 2); piece(kt + 3, 3); }
            __builtin_amdgcn_sched_barrier(0);
            if (two) {
#pragma unroll
                for (int j = 0; j < 2; ++j) frag(st1 + fb + j * 128, ch[j], cl[j]);
                frag(st1 + fa, ah[0], al[0]);
            }
            mma_group(std::integral_constant<int, 3>{}, ah[1], al[1], bh, bl);
            __builtin_amdgcn_sched_barrier(0);
            if (two) {
                frag(st1 + fa + 128, ah[1], al[1]);
                mma_group(std::integral_constant<int, 0>{}, ah[0], al[0], ch, cl);
                __builtin_amdgcn_sched_barrier(0);
                frag(st1 + fa + 256, ah[0], al[0]);
                mma_group(std::integral_constant<int, 1>{}, ah[1], al[1], ch, cl);
                __builtin_amdgcn_sched_barrier(0);
                frag(st1 + fa + 384, ah[1], al[1]);
                mma_group(std::integral_constant<int, 2>{}, ah[0], al[0], ch, cl);
                __builtin_amdgcn_sched_barrier(0);
                mma_group(std::integral_constant<int, 3>{}, ah[1], al[1], ch, cl);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef XPS_DMA_STAMP
            DSTAMP(t4)
            s_wait += t1 - t0; s_bar += t2 - t1; s_dma += t3 - t2; s_mma += t4 - t3;
#endif
        }
    } else
    for (int kt = 0; kt < nkt; ++kt) {
        DSTAMP(t0)
        wait_tile();
        DSTAMP(t1)
        __builtin_amdgcn_s_barrier();                          // ... everybody's; and everybody is done reading k-tile kt - 1
        DSTAMP(t2)
        if (!XPS_DMA_SPREAD) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) piece(kt + DMA_D - 1, pc);
        }
        DSTAMP(t3)
        const unsigned char* st = smem + (kt % DMA_D) * DMA_STAGE;
        bf16x8 bh[2], bl[2], ah[2], al[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) frag(st + fb + j * 128, bh[j], bl[j]);
        frag(st + fa, ah[0], al[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (XPS_DMA_SPREAD) { piece(kt + DMA_D - 1, 0); piece(kt + DMA_D - 1, 1); }
        __builtin_amdgcn_sched_barrier(0);
        auto group = [&](auto I) {
            constexpr int i = decltype(I)::value;
            if (i + 1 < 4) frag(st + fa + (i + 1) * 128, ah[(i + 1) & 1], al[(i + 1) & 1]);
            mma_group(I, ah[i & 1], al[i & 1], bh, bl);
            __builtin_amdgcn_sched_barrier(0);
            if (XPS_DMA_SPREAD && i < 2) {
                piece(kt + DMA_D - 1, 2 + i);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        group(std::integral_constant<int, 0>{});
        group(std::integral_constant<int, 1>{});
        group(std::integral_constant<int, 2>{});
        group(std::integral_constant<int, 3>{});
#ifdef XPS_DMA_STAMP
        DSTAMP(t4)
        s_wait += t1 - t0; s_bar += t2 - t1; s_dma += t3 - t2; s_mma += t4 - t3;
#endif
    }
#ifdef XPS_DMA_STAMP
    if (lane == 0) {
        unsigned long long* o = g_dma_stamp + ((blockIdx.x * 8 + wave) & 4095) * 4;
        o[0] = s_wait; o[1] = s_bar; o[2] = s_dma; o[3] = s_mma;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the re-read tail pieces: nothing may land after the hand-back
    __builtin_amdgcn_s_barrier();
}

// the 256 column sums of a tile (tn_dma_pipeline<true>): wave w holds those of columns 128 (w / 4) + 32 (w % 4) + 0..31 of the
// tile in column 0 of cacc (C/D layout of the 32 x 32 MFMA: lanes 0 and 32, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
__device__ inline void dma_colsum_store(const f32x16& cacc, float* __restrict__ dst, int cs_lo = 0, int cs_hi = 8) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cgrp = (wave >> 2) * 4 + (wave & 3);
    if ((lane & 31) == 0 && cgrp >= cs_lo && cgrp < cs_hi) {
        float* d = dst + (wave >> 2) * 128 + (wave & 3) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) d[(r & 3) + 8 * (r >> 2)] = cacc[r];
    }
}



// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA k loop for the DIRECT forms (C = A B^T: NT, both operands stored [x][k]; C = A B: NN, A [x][k], B [k][x]) on
// XPS_FMT_SPLIT4 operands, 32-deep stages.
//
// [x][k] operand ("KC"): a row holds its k values contiguously, 32 of them = 8 split4 groups = 128 bytes = ONE cache line.  A
// DMA piece moves 8 rows x 128 B (full lines); the image is the 256 rows back to back (32 KB, no padding).  An MFMA fragment
// (lane -> row, 8 consecutive k) needs the hi halves of TWO neighbouring groups: each group is read whole (ds_read_b128:
// hi[0..3] | lo[0..3]) and the halves of the pair are regrouped in registers (the alternative, ds_read2_b64 of the hi halves
// only, runs at half the LDS rate and cannot be made conflict-free: hi halves occupy every other 8-byte bank pair).  Bank
// conflicts: rows are 128 B apart, so the 16 rows of a b128 lane group would share two 16-byte slots; the 16-byte group g of
// row r therefore sits at group position g ^ ((r >> 1) & 7) -- applied on the SOURCE side (each lane of a piece fetches the
// group that belongs at its linear LDS position: a permutation inside one cache line) and undone by the fragment addresses.
// [k][x] operand ("KX", the B of the NN form): two 16-row images of the weight-gradient loop above (transposing reads).
//
// Two stages (2 x ~64 KB): stage s + 1 is in flight while stage s is multiplied; its refill is issued behind the barrier that
// ends the multiplication.  Per accumulator the MFMAs come k-tile by k-tile (lo*hi, hi*lo, hi*hi): same bits as every other
// tile kernel.
constexpr int KC_IMG = 256 * 128;                                  // 32768 B
template <bool BKX> struct DirectStage { static constexpr int B_IMG = BKX ? 2 * DMA_IMG : KC_IMG; static constexpr int BYTES = KC_IMG + B_IMG; };
template <bool BKX> constexpr int direct_dma_lds() { return 2 * DirectStage<BKX>::BYTES; }

// acc += A[m0.., kbeg .. kbeg + 32 nks) op(B)[.., n0..]; A: [m][k] split4 (lda elements per row); B: BKX ? [k][n] : [n][k], split4
template <bool BKX>
__device__ inline void direct_dma_pipeline(f32x16 (&acc)[4][2], const float* __restrict__ A, long long lda, const float* __restrict__ B,
                                           long long ldb, int m0, int n0, int kbeg, int nks, unsigned char* smem, const bool active = true) {
    // active (wave-uniform): false = this wave's 64 output columns are beyond the problem (skinny N in a 256-wide tile): it moves
    // its DMA pieces and keeps the barriers, but reads no fragments and issues no MFMAs
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    constexpr int STAGE = DirectStage<BKX>::BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    const unsigned lds0 = dma_lds_base(smem);
    if (nks <= 0) return;
    // ---- DMA roles.  KC image: wave w moves the pieces w, w + 8, w + 16, w + 24 (rows 8 p .. 8 p + 7): lane -> (row 8 p + lane / 8,
    //      group (lane % 8) ^ key(row)); key = ((row >> 1) & 7) is the same for the wave's four pieces (64 rows apart)
    const int rr = lane >> 3, cpos = lane & 7;
    const int keyw = ((8 * wave + rr) >> 1) & 7;
    const unsigned va = (unsigned)((long long)(8 * wave + rr) * lda * 4 + ((cpos ^ keyw) * 16));
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(A + (long long)m0 * lda + kbeg);
    const long long pstepa = 64 * lda * 4;                          // next piece of the wave: 64 rows further
    unsigned vb, vb1 = 0;
    const unsigned char* bbase;
    long long pstepb, kstepb;
    if (BKX) {
        // KX image: wave w moves rows 2 w, 2 w + 1 of both 16-row k-tiles
        vb = (unsigned)((long long)(2 * wave) * ldb * 4 + lane * 16);
        vb1 = vb + (unsigned)(ldb * 4);
        bbase = reinterpret_cast<const unsigned char*>(B + (long long)kbeg * ldb + n0);
        pstepb = 16 * ldb * 4;
        kstepb = 32 * ldb * 4;
    } else {
        vb = (unsigned)((long long)(8 * wave + rr) * ldb * 4 + ((cpos ^ keyw) * 16));
        bbase = reinterpret_cast<const unsigned char*>(B + (long long)n0 * ldb + kbeg);
        pstepb = 64 * ldb * 4;
        kstepb = 128;
    }
    const unsigned ra0 = (unsigned)dma_rowoff(2 * wave), ra1 = (unsigned)dma_rowoff(2 * wave + 1);
    // piece pc (0..3: A, 4..7: B) of stage ks (beyond the range: the last stage again, into a buffer nobody reads)
    auto piece = [&](int ks, int pc) {
        const int kk = ks < nks ? ks : nks - 1;
        const unsigned st = lds0 + (unsigned)(ks & 1) * STAGE;
        if (pc < 4) dma_piece(abase + (long long)kk * 128 + pc * pstepa, va, st + (unsigned)(wave + 8 * pc) * 1024u);
        else if (!BKX) dma_piece(bbase + (long long)kk * kstepb + (pc - 4) * pstepb, vb, st + KC_IMG + (unsigned)(wave + 8 * (pc - 4)) * 1024u);
        else {
            const int t = (pc - 4) >> 1;                               // k-tile of the stage; (pc & 1): row 2 w / 2 w + 1
            dma_piece(bbase + (long long)kk * kstepb + t * pstepb, (pc & 1) ? vb1 : vb, st + KC_IMG + (unsigned)t * DMA_IMG + ((pc & 1) ? ra1 : ra0));
        }
    };
    // ---- fragment addresses.  KC: lane -> row x0 + (lane & 31), k = 16 t + 8 h + 0..7 = groups 4 t + 2 h, + 1 at positions ^ key
    const int r = lane & 31, h = lane >> 5, key = (r >> 1) & 7;
    int fka[2][2];                                                    // [t][j]: byte offset inside the image for row group 0 of the wave
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) fka[t][j] = r * 128 + (((4 * t + 2 * h + j) ^ key) << 4);
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int fro = dma_rowoff(8 * (g >> 1) + q) + ((g & 1) * 4 + pp) * 16;       // KX fragment offset (see tn_dma_pipeline)
    auto frag_kc = [&](const unsigned char* img, int x0, int t, bf16x8& fh, bf16x8& fl) {
        const bf16x8 ga = *reinterpret_cast<const bf16x8*>(img + x0 * 128 + fka[t][0]);
        const bf16x8 gb = *reinterpret_cast<const bf16x8*>(img + x0 * 128 + fka[t][1]);
        fh = __builtin_shufflevector(ga, gb, 0, 1, 2, 3, 8, 9, 10, 11);
        fl = __builtin_shufflevector(ga, gb, 4, 5, 6, 7, 12, 13, 14, 15);
    };
    auto frag_kx = [&](const unsigned char* img, int x0, int t, bf16x8& fh, bf16x8& fl) {
        const unsigned char* p = img + t * DMA_IMG + fro + (x0 / 4) * 16;
        const bf16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
        const bf16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + DMA_ROW4));
        const bf16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 8));
        const bf16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + DMA_ROW4 + 8));
        fh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        fl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
#ifndef XPS_DIRECT_MERGED
#define XPS_DIRECT_MERGED 0
#endif
#if XPS_DIRECT_MERGED
    // (Built and measured, not the default: NT 40960-row projection shape 299.5 / 303.0 us against 299.2 / 299.5 us for the
    //  two-barrier form below, NN 279.9 / 279.7 against 283.1 / 286.5 us -- profiles/round4: the compiler already rotates the
    //  two-barrier loop so that the refill burst sits among the last MFMAs of a stage, and the refill keeps a whole stage of lead.)
    // ONE barrier per stage: "every piece of stage ks has landed" and "everybody is done reading stage ks - 1" are the same
    // barrier when the refill of the freed buffer (stage ks + 1) is issued BEHIND it, spread over the MFMA groups of the first
    // k-tile (two pieces per row group) -- stamps of the first form (refill of stage ks + 2 in one burst behind a second barrier):
    // per 32-deep stage and wave 44 cycles waiting for data, 1199 in the two barriers, 697 issuing the burst with an idle matrix
    // pipe, 2974 in reads + MFMAs (3072 is the floor per SIMD).
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) piece(0, pc);
    for (int ks = 0; ks < nks; ++ks) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of stage ks (the only ones in flight) have landed
        __builtin_amdgcn_s_barrier();                              // ... everybody's; and everybody is done reading stage ks - 1
        const unsigned char* sa = smem + (ks & 1) * STAGE;
        const unsigned char* sb = sa + KC_IMG;
        const bool more = ks + 1 < nks;                            // (no refill behind the last stage: nothing may land after the hand-back)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 bh[2], bl[2];
            if (active) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (BKX) frag_kx(sb, wn + 32 * j, t, bh[j], bl[j]);
                    else frag_kc(sb, wn + 32 * j, t, bh[j], bl[j]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (active) {
                    bf16x8 ah, al;
                    frag_kc(sa, wm + 32 * i, t, ah, al);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
                    }
                }
                if (t == 0 && more) {
                    __builtin_amdgcn_sched_barrier(0);
                    piece(ks + 1, 2 * i); piece(ks + 1, 2 * i + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#else
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) piece(0, pc);
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) piece(1, pc);
#ifdef XPS_DMA_STAMP
    unsigned long long u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0, u5 = 0, z_wait = 0, z_bar = 0, z_mma = 0, z_bar2 = 0, z_dma = 0;
#endif
    for (int ks = 0; ks < nks; ++ks) {
        DSTAMP(u0)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // this wave's pieces of stage ks have landed (stage ks + 1 in flight)
        DSTAMP(u1)
        __builtin_amdgcn_s_barrier();
        DSTAMP(u2)
        const unsigned char* sa = smem + (ks & 1) * STAGE;
        const unsigned char* sb = sa + KC_IMG;
        if (active)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (BKX) frag_kx(sb, wn + 32 * j, t, bh[j], bl[j]);
                else frag_kc(sb, wn + 32 * j, t, bh[j], bl[j]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x8 ah, al;
                frag_kc(sa, wm + 32 * i, t, ah, al);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DSTAMP(u3)
        __builtin_amdgcn_s_barrier();                               // everybody is done reading stage ks: refill its buffer
        DSTAMP(u4)
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) piece(ks + 2, pc);
#ifdef XPS_DMA_STAMP
        DSTAMP(u5)
        z_wait += u1 - u0; z_bar += u2 - u1; z_mma += u3 - u2; z_bar2 += u4 - u3; z_dma += u5 - u4;
#endif
    }
#ifdef XPS_DMA_STAMP
    if (lane == 0) {
        unsigned long long* o = g_dma_stamp + ((blockIdx.x * 8 + wave) & 4095) * 4;
        o[0] = z_wait; o[1] = z_bar + z_bar2; o[2] = z_dma; o[3] = z_mma;
    }
#endif
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

}  // namespace xps_big
