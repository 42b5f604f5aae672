// 256 x 256 output tiles on 8 waves (512 threads): the bf16 split-product pipeline of xps_gemm_tile.h for LARGE interior
// shapes (the configs[3] layer GEMMs: 40960 x 1536 x 1024 projections, 40960 x 1024 x 3072 input gradients, 1536 x 1024 x
// 40960 weight gradients).  Same arithmetic as the 128 x 128 kernels -- operands split hi + lo while they are staged, per
// 32 x 32 accumulator tile and k-tile the products lo*hi, hi*lo, hi*hi in that order -- so a product whose k range is not
// split differently has the same bits as from the small-tile kernels.
//
// Why a second tile shape: the split costs ~12 issue cycles per staged element and a wave has 24 free issue cycles per
// v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md, cycle constants).  Staged elements per MFMA go with (BM + BN) / (BM BN),
// fragment reads per MFMA with (WM + WN) / (WM WN): 128 x 128 blocks of 64 x 64 wave tiles spend ~27 issue cycles per
// MFMA beside the MFMA itself (issue-bound: 0.30 of the bf16 peak measured on 8192^3), 256 x 256 blocks of 128 x 64 wave
// tiles ~15.  One block per CU (96 KB of LDS, 2 waves per SIMD at <= 256 registers).
//
// Preconditions (checked on the host, see big_ok): M % 256 == 0, N % 256 == 0, every k range a multiple of 16, plain
// leading dimensions (no row groups), 16-byte aligned operands.
#pragma once
#include "xps_gemm_tile.h"

namespace xps_big {
using namespace xps_tile;

constexpr int TM = 256, TN = 256, NTHR = 512;

constexpr int BIG_NS = 2;
struct BigStage {
    BfTile<256> a[BIG_NS];
    BfTile<256> b[BIG_NS];
};

// thread -> two 16-byte vectors of a 256 x 16 k-tile.  KC ([x][k]): row x = tid / 4 + 128 r, k = 4 (tid % 4);
// !KC ([k][x]): k row = tid / 64 + 8 r (one wave reads one 1-KiB row), x = 4 (tid % 64)
template <bool KC>
struct BigLoader {
    const float* p[2];
    long long kstep;
    __device__ inline void init(const float* __restrict__ P, long long ld, int x0, int k0, int tid) {
        if (KC) {
#pragma unroll
            for (int r = 0; r < 2; ++r) p[r] = P + (long long)(x0 + (tid >> 2) + 128 * r) * ld + k0 + (tid & 3) * 4;
            kstep = BKT;
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r) p[r] = P + (long long)(k0 + (tid >> 6) + 8 * r) * ld + x0 + (tid & 63) * 4;
            kstep = (long long)BKT * ld;
        }
    }
    __device__ inline void load(f32x4 (&v)[2], int kt) const {
#pragma unroll
        for (int r = 0; r < 2; ++r) v[r] = *reinterpret_cast<const f32x4*>(p[r] + kt * kstep);
    }
};

template <bool KC>
__device__ inline void big_stage(const f32x4 (&v)[2], BfTile<256>& S, int tid, const bool pre) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        bf16x4 h, l;
        stage_split(v[r], pre, h, l);
        if (KC) {
            const int x = (tid >> 2) + 128 * r, k4 = (tid & 3) * 4;
            *reinterpret_cast<bf16x4*>(&S.hi[x][k4]) = h;
            *reinterpret_cast<bf16x4*>(&S.lo[x][k4]) = l;
        } else {
            const int k = (tid >> 6) + 8 * r, x4 = (tid & 63) * 4;
            *reinterpret_cast<bf16x4*>(&S.thi[k][x4]) = h;
            *reinterpret_cast<bf16x4*>(&S.tlo[k][x4]) = l;
        }
    }
}

// Accumulates nkt k-tiles.  acc[i][j]: rows wm + 32 i, columns wn + 32 j of the block tile, wave w -> wm = 128 (w / 4),
// wn = 64 (w % 4); C/D layout of the 32 x 32 MFMA.  csum (want_csum, [k][x] A operands only): running sums of this
// thread's 4 x columns over its k rows (the bias gradient of the weight-gradient form), taken from the staging registers.
// PA / PB: the operand is XPS_FMT_SPLIT4 (compile-time: a run-time flag would put a branch into the k loop and keep the
// scheduler from interleaving the staging with the MFMAs)
template <bool AK, bool BK, bool PA, bool PB>
__device__ inline void big_pipeline_t(f32x16 (&acc)[4][2], f32x4& csum, const bool want_csum, const BigLoader<AK>& la,
                                      const BigLoader<BK>& lb, const int nkt, BigStage& S) {
    constexpr bool preA = PA, preB = PB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    f32x4 ra0[2], rb0[2], ra1[2], rb1[2];                 // k-tiles kt + 1 and kt + 2 in flight
    if (nkt > 0) {
        la.load(ra0, 0); lb.load(rb0, 0);
        if (nkt > 1) { la.load(ra1, 1); lb.load(rb1, 1); }
        big_stage<AK>(ra0, S.a[0], tid, preA);
        big_stage<BK>(rb0, S.b[0], tid, preB);
        if (!AK && want_csum) csum += stage_values(ra0[0], preA) + stage_values(ra0[1], preA);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
#ifndef XPS_BIG_ABL_NOLOAD        // (timing ablation, wrong results: the k loop without its global loads)
        if (kt + 2 < nkt) { la.load(ra0, kt + 2); lb.load(rb0, kt + 2); }
#endif
        {
#ifndef XPS_BIG_PIN      // (pinned A-fragment prefetch measured: within 1 % on every shape of tools/bench_gemm_h512.py -- not the default)
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf_frag<BK, 256>(S.b[buf], wn + j * 32, lane, bh[j], bl[j]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x8 ah, al;
                bf_frag<AK, 256>(S.a[buf], wm + i * 32, lane, ah, al);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
                }
            }
#else
            // The A fragment of row group i + 1 is requested BEFORE the six MFMAs of row group i (two register sets, pinned with
            // scheduling barriers): left alone the scheduler requests it after four of them and waits for it behind the sixth -- two
            // MFMAs (64 cycles) of cover for an LDS read, four times per k-tile on both waves of a SIMD at once.
            bf16x8 bh[2], bl[2], ah[2], al[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf_frag<BK, 256>(S.b[buf], wn + j * 32, lane, bh[j], bl[j]);
            bf_frag<AK, 256>(S.a[buf], wm, lane, ah[0], al[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i + 1 < 4) bf_frag<AK, 256>(S.a[buf], wm + (i + 1) * 32, lane, ah[(i + 1) & 1], al[(i + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i & 1], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i & 1], bh[j], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
        if (kt + 1 < nkt) {
#ifndef XPS_BIG_ABL_NOSTAGE       // (timing ablation, wrong results: no LDS stores of the next k-tile)
            big_stage<AK>(ra1, S.a[buf ^ 1], tid, preA);
            big_stage<BK>(rb1, S.b[buf ^ 1], tid, preB);
#endif
            if (!AK && want_csum) csum += stage_values(ra1[0], preA) + stage_values(ra1[1], preA);
#pragma unroll
            for (int r = 0; r < 2; ++r) { ra1[r] = ra0[r]; rb1[r] = rb0[r]; }
        }
#ifndef XPS_BIG_ABL_NOBARRIER      // (timing ablation, wrong results: no barrier per k-tile)
        __syncthreads();
#endif
    }
}

template <bool AK, bool BK>
__device__ inline void big_pipeline(f32x16 (&acc)[4][2], f32x4& csum, const bool want_csum, const BigLoader<AK>& la,
                                    const BigLoader<BK>& lb, const int nkt, BigStage& S, const bool preA, const bool preB) {
    if (preA) {
        if (preB) big_pipeline_t<AK, BK, true, true>(acc, csum, want_csum, la, lb, nkt, S);
        else big_pipeline_t<AK, BK, true, false>(acc, csum, want_csum, la, lb, nkt, S);
    } else {
        if (preB) big_pipeline_t<AK, BK, false, true>(acc, csum, want_csum, la, lb, nkt, S);
        else big_pipeline_t<AK, BK, false, false>(acc, csum, want_csum, la, lb, nkt, S);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 32-deep stages (k ranges that are multiples of 32).  What it changes against the 16-deep pipeline above: [x][k]
// operands are fetched in FULL 128-byte lines (8 lanes x 16 B per row; 16-deep k-tiles took half a line per visit and the
// other half one k-tile later, from L2 again), and a barrier guards 48 MFMAs per wave instead of 24.  LDS: an [x][k] operand
// keeps its two 16-deep k-tiles as two swizzled images without padding (hi / lo: 256 rows x 32 B; 16-byte chunk kh of row r
// sits at chunk kh ^ ((r >> 3) & 1): the b128 fragment reads of a 16-lane group {0-3, 12-15, 20-27}, ... touch 16 distinct
// 4-bank groups; the second k-tile's image starts 64 B late so that the 8-byte stores of a row's two k-tiles use different
// banks); a [k][x] operand keeps 32 k rows of 576 B as before.  One register set: the k-stage after the current one is
// requested at the top of an iteration and split + stored behind its MFMAs.
struct KcImage {                                   // [x][k] operand, one 32-deep stage
    __bf16 hi0[256][16];
    __bf16 pad0[32];
    __bf16 hi1[256][16];
    __bf16 pad1[32];
    __bf16 lo0[256][16];
    __bf16 pad2[32];
    __bf16 lo1[256][16];
    __bf16 pad3[32];
};
struct KxImage {                                   // [k][x] operand, one 32-deep stage
    __bf16 thi[32][256 + 32];
    __bf16 tlo[32][256 + 32];
};
union Image32 {
    KcImage kc;
    KxImage kx;
};
struct BigStage32 {
    Image32 a[2];
    Image32 b[2];
};

template <bool KC>
struct BigLoader32 {
    const float* p[4];
    long long kstep;
    __device__ inline void init(const float* __restrict__ P, long long ld, int x0, int k0, int tid) {
        if (KC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = P + (long long)(x0 + (tid >> 3) + 64 * r) * ld + k0 + (tid & 7) * 4;
            kstep = 32;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) p[r] = P + (long long)(k0 + (tid >> 6) + 8 * r) * ld + x0 + (tid & 63) * 4;
            kstep = 32 * ld;
        }
    }
    __device__ inline void load(f32x4 (&v)[4], int ks) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = *reinterpret_cast<const f32x4*>(p[r] + ks * kstep);
    }
};

template <bool KC>
__device__ inline void big_stage32(const f32x4 (&v)[4], Image32& S, int tid, const bool pre) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bf16x4 h, l;
        stage_split(v[r], pre, h, l);
        if (KC) {
            const int row = (tid >> 3) + 64 * r, kc = tid & 7, c4 = kc & 3;
            const int col = (((c4 >> 1) ^ ((row >> 3) & 1)) << 3) + ((c4 & 1) << 2);
            if (kc < 4) {
                *reinterpret_cast<bf16x4*>(&S.kc.hi0[row][col]) = h;
                *reinterpret_cast<bf16x4*>(&S.kc.lo0[row][col]) = l;
            } else {
                *reinterpret_cast<bf16x4*>(&S.kc.hi1[row][col]) = h;
                *reinterpret_cast<bf16x4*>(&S.kc.lo1[row][col]) = l;
            }
        } else {
            const int k = (tid >> 6) + 8 * r, x4 = (tid & 63) * 4;
            *reinterpret_cast<bf16x4*>(&S.kx.thi[k][x4]) = h;
            *reinterpret_cast<bf16x4*>(&S.kx.tlo[k][x4]) = l;
        }
    }
}

// MFMA operand fragment of k-tile t (0 / 1) of a 32-deep stage: rows x0 .. x0 + 31
template <bool KC>
__device__ inline void big_frag32(const Image32& S, int t, int x0, int lane, bf16x8& fh, bf16x8& fl) {
    if constexpr (KC) {
        const int row = x0 + (lane & 31), col = (((lane >> 5) ^ ((row >> 3) & 1)) << 3);
        if (t == 0) {
            fh = *reinterpret_cast<const bf16x8*>(&S.kc.hi0[row][col]);
            fl = *reinterpret_cast<const bf16x8*>(&S.kc.lo0[row][col]);
        } else {
            fh = *reinterpret_cast<const bf16x8*>(&S.kc.hi1[row][col]);
            fl = *reinterpret_cast<const bf16x8*>(&S.kc.lo1[row][col]);
        }
    } else {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
        const int k0 = 16 * t + 8 * (g >> 1) + q, c0 = x0 + (g & 1) * 16 + 4 * pp;
        const bf16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(&S.kx.thi[k0][c0]));
        const bf16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(&S.kx.thi[k0 + 4][c0]));
        const bf16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(&S.kx.tlo[k0][c0]));
        const bf16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(&S.kx.tlo[k0 + 4][c0]));
        fh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        fl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

template <bool AK, bool BK>
__device__ inline void big_mma32(f32x16 (&acc)[4][2], const BigStage32& S, int buf) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) big_frag32<BK>(S.b[buf], t, wn + j * 32, lane, bh[j], bl[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x8 ah, al;
            big_frag32<AK>(S.a[buf], t, wm + i * 32, lane, ah, al);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
            }
        }
    }
}

// nks 32-deep k-stages; same accumulation order per accumulator as the 16-deep pipelines (k-tile by k-tile)
template <bool AK, bool BK>
__device__ inline void big_pipeline32(f32x16 (&acc)[4][2], f32x4& csum, const bool want_csum, const BigLoader32<AK>& la,
                                      const BigLoader32<BK>& lb, const int nks, BigStage32& S, const bool preA, const bool preB) {
    const int tid = threadIdx.x;
    if (nks <= 0) return;
    f32x4 va[4], vb[4];
    la.load(va, 0); lb.load(vb, 0);
    big_stage32<AK>(va, S.a[0], tid, preA);
    big_stage32<BK>(vb, S.b[0], tid, preB);
    if (!AK && want_csum) csum += (stage_values(va[0], preA) + stage_values(va[1], preA)) + (stage_values(va[2], preA) + stage_values(va[3], preA));
    __syncthreads();
    int ks = 0;
    for (; ks + 1 < nks; ++ks) {
        const int buf = ks & 1;
        la.load(va, ks + 1); lb.load(vb, ks + 1);
        __builtin_amdgcn_sched_barrier(0);             // requests first: a whole stage of MFMAs hides them
        big_mma32<AK, BK>(acc, S, buf);
        big_stage32<AK>(va, S.a[buf ^ 1], tid, preA);
        big_stage32<BK>(vb, S.b[buf ^ 1], tid, preB);
        if (!AK && want_csum) csum += (stage_values(va[0], preA) + stage_values(va[1], preA)) + (stage_values(va[2], preA) + stage_values(va[3], preA));
        __syncthreads();
    }
    big_mma32<AK, BK>(acc, S, ks & 1);
    __syncthreads();
}

__device__ inline void big_zero(f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// C tile store (plain leading dimension): a store instruction writes 2 rows x 32 consecutive floats per wave
__device__ inline void big_store_c(const f32x16 (&acc)[4][2], float* __restrict__ C, long long ldc, const float* __restrict__ bias,
                                   int m0, int n0, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    const int li = lane & 31, lk = lane >> 5;
    float* cbase = C + (long long)(m0 + wm + 4 * lk) * ldc + n0 + wn + li;
    float bv[2] = {0.f, 0.f};
    if (bias) { bv[0] = bias[n0 + wn + li]; bv[1] = bias[n0 + wn + 32 + li]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* crow = cbase + (long long)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v = acc[i][j][r] + bv[j];
                if (accumulate) v += crow[j * 32];
                __builtin_nontemporal_store(v, &crow[j * 32]);
            }
        }
}

// the same for a tile whose columns >= nvalid do not exist (a skinny problem, N < 256, in ONE 256-wide tile: C has N columns)
__device__ inline void big_store_c_masked(const f32x16 (&acc)[4][2], float* __restrict__ C, long long ldc, const float* __restrict__ bias,
                                          int m0, int nvalid, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
    const int li = lane & 31, lk = lane >> 5;
    if (wn >= nvalid) return;
    float* cbase = C + (long long)(m0 + wm + 4 * lk) * ldc + wn + li;
    const bool ok[2] = {wn + li < nvalid, wn + 32 + li < nvalid};
    float bv[2] = {0.f, 0.f};
    if (bias) { bv[0] = ok[0] ? bias[wn + li] : 0.f; bv[1] = ok[1] ? bias[wn + 32 + li] : 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* crow = cbase + (long long)(i * 32 + (r & 3) + 8 * (r >> 2)) * ldc;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!ok[j]) continue;
                float v = acc[i][j][r] + bv[j];
                if (accumulate) v += crow[j * 32];
                __builtin_nontemporal_store(v, &crow[j * 32]);
            }
        }
}

// Split-K slab of a 256 x 256 tile, written as its four 128 x 128 sub-tiles in the register order of the small-tile
// kernels (xps_gemm.hip: slab_store / slab_decode), so that ONE reduce kernel serves both tile shapes: sub-tile
// (wave / 4, (wave % 4) / 2), its "wave" = 2 (i / 2) + wave % 2, its accumulator tile (i % 2, j).
// sub[] = the four sub-tiles' slab addresses (row-major 2 x 2).
__device__ inline void big_slab_store(const f32x16 (&acc)[4][2], float* const (&sub)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* base = sub[(wave >> 2) * 2 + ((wave & 3) >> 1)] + lane * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                const int wave_o = (i >> 1) * 2 + (wave & 1);
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base + wave_o * 4096 + (((i & 1) * 2 + j) * 4 + q) * 256));
            }
}

}  // namespace xps_big
