// Shared fp32-MFMA tile machinery (v_mfma_f32_32x32x2_f32): k-major LDS staging, two-k-tile-deep register
// prefetch, 64- or 128-row output tiles.  Used by the GEMM entry points (xps_gemm.hip) and by the per-step
// fused GRU kernels for large hidden sizes (xps_gru.hip).
#pragma once
#include "xps_common.h"

namespace xps_tile {

#ifndef XPS_BKT
#define XPS_BKT 16
#endif
#ifndef XPS_GEMM_WAVES
#define XPS_GEMM_WAVES 1
#endif
constexpr int BM = 128, BN = 128, BKT = XPS_BKT, LDT = 132;

#ifdef XPS_GSTAMP
// diagnostic build only: per-wave cycle shares of the k loop (written to a buffer no kernel reads)
static __device__ unsigned long long g_gstamp[8192 * 4];
#define GSTAMP(var)                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");      \
    __builtin_amdgcn_sched_barrier(0);
#else
#define GSTAMP(var)
#endif

constexpr int KL = BKT / 4;          // KCONTIG: lanes covering one row's k range

// XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (each with a private 4 MB L2), so
// blocks b, b+8, b+16, ... share an L2.  The remap gives every XCD a CONTIGUOUS range of logical ids
// (bijective for any grid size); kernels then order logical ids so that neighbours read the same operand
// panel -> a panel is fetched from HBM / Infinity Cache once per XCD instead of once per block.
__device__ inline int xcd_remap(int bid, int nwg) {
    constexpr int NX = 8;
    const int x = bid % NX, q = nwg / NX, r = nwg % NX;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + bid / NX;
}

// Stages a (BKT x W) k-major tile of a matrix stored either [x][k] (KCONTIG) or [k][x]; W = 64 or 128.
// BF: tag of the loaders used by the bf16 split-product pipeline (gemm_accumulate_bf); same geometry.
template <bool KCONTIG, int W, bool BF = false>
struct TileLoader {
    // KCONTIG : 16-byte vectors along k; thread -> (x = tid / KL (+ XR per pass), k4 = (tid % KL) * 4)
    // !KCONTIG: 16-byte vectors along x; thread -> (k = tid / XL (+ KR per pass), x4 = (tid % XL) * 4)
    // `fast` (block-uniform): the W-wide x range is fully inside the matrix, vectors are aligned and
    // the row map is a plain leading dimension -> no per-element guards, no divisions in the k loop.
    static constexpr int XR = 256 / KL;                 // KCONTIG: x rows per pass
    static constexpr int XL = W / 4;                    // !KCONTIG: lanes per k row
    static constexpr int KR = 256 / XL;                 // !KCONTIG: k rows per pass
    static constexpr int NV = KCONTIG ? W / XR : BKT / KR;
    // !KCONTIG thread map: k row of register r, 4-wide x group of the thread (a k row is read by XL adjacent lanes:
    // W * 4 contiguous bytes).  Both pipelines use it: the bf16 one keeps [k][x] operands un-transposed in LDS and reads
    // the MFMA fragments with the transposing ds_read_b64_tr_b16 (bf_frag).
    __device__ static inline int krow(int tid, int r) { return tid / XL + KR * r; }
    __device__ static inline int xgrp(int tid) { return tid % XL; }
    const float* base[NV];
    long long xoff[NV];
    long long kstride, kgs;       // !KCONTIG: row stride inside a group / group stride of the k rows
    int krpg;                     // !KCONTIG: k rows per group (0: plain leading dimension)
    bool fast;                    // unguarded 16-byte loads are possible (alignment, row-map structure)
    bool full;                    // ... and the W-wide x range lies entirely inside the matrix
    bool ok[NV];                  // edge tiles: this thread's vector lies inside the matrix (else it reads as 0)

    __device__ inline void init(const float* __restrict__ P, const RowMap& rm, int x0, int X, int rows_k, int tid, bool vec) {
        if (KCONTIG) {
            // edge tiles (x0 + W > X) stay on the fast path: rows beyond the matrix are predicated off and read as 0
            fast = vec;
            full = x0 + W <= X;
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int x = x0 + tid / KL + XR * r;
                xoff[r] = (x < X) ? rm.off(x) : -1;
                ok[r] = x < X;
                base[r] = P + (xoff[r] >= 0 ? xoff[r] : 0) + (tid % KL) * 4;
            }
            kstride = 1;
        } else {
            // k rows under a two-level map stay on the fast path when a k-tile never straddles a group
            // (rows per group a multiple of the tile depth; k-split chunks start on tile boundaries)
            const bool plain = rm.rpg >= rows_k;
            // edge tiles stay fast when whole 16-byte vectors are in or out (X % 4 == 0): out-of-range lanes read 0
            full = x0 + W <= X;
            fast = vec && (full || X % 4 == 0) && (plain || rm.rpg % BKT == 0);
            const bool lane_in = x0 + xgrp(tid) * 4 < X;
#pragma unroll
            for (int r = 0; r < NV; ++r) ok[r] = lane_in;
            kstride = rm.ld;
            kgs = rm.gs;
            krpg = plain ? 0 : rm.rpg;
#pragma unroll
            for (int r = 0; r < NV; ++r)
                base[r] = P + (long long)krow(tid, r) * rm.ld + (lane_in ? x0 + xgrp(tid) * 4 : 0);
        }
    }

    // unguarded 16-byte loads of a FULL k-tile (only valid when `fast`; EDGE = false additionally needs `full`).
    // EDGE: vectors outside the matrix are predicated off and read as zero.
    template <bool EDGE>
    __device__ inline void load_fast(f32x4 (&v)[NV], int kt0) const {
        long long koff;                                   // block-uniform offset of the tile's first k row
        if (KCONTIG) koff = kt0;
        else if (krpg == 0) koff = (long long)kt0 * kstride;
        else koff = (long long)(kt0 / krpg) * kgs + (long long)(kt0 % krpg) * kstride;
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            if (EDGE) v[r] = ok[r] ? *reinterpret_cast<const f32x4*>(base[r] + koff) : f32x4{0.f, 0.f, 0.f, 0.f};
            else v[r] = *reinterpret_cast<const f32x4*>(base[r] + koff);
        }
    }

    __device__ inline void load(f32x4 (&v)[NV], const float* __restrict__ P, const RowMap& rm, int x0, int X,
                                int kt0, int kend, int tid, bool vec) const {
        if (fast && full && kt0 + BKT <= kend) {
            load_fast<false>(v, kt0);
            return;
        }
        if (KCONTIG) {
            const int k = kt0 + (tid % KL) * 4;
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
                if (xoff[r] >= 0) {
                    const float* p = P + xoff[r] + k;
                    if (vec && k + 3 < kend) {
                        t = *reinterpret_cast<const f32x4*>(p);
                    } else {
                        if (k + 0 < kend) t.x = p[0];
                        if (k + 1 < kend) t.y = p[1];
                        if (k + 2 < kend) t.z = p[2];
                        if (k + 3 < kend) t.w = p[3];
                    }
                }
                v[r] = t;
            }
        } else {
            const int x = x0 + xgrp(tid) * 4;
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int k = kt0 + krow(tid, r);
                f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
                if (k < kend && x < X) {
                    const float* p = P + rm.off(k) + x;
                    if (vec && x + 3 < X) {
                        t = *reinterpret_cast<const f32x4*>(p);
                    } else {
                        t.x = p[0];
                        if (x + 1 < X) t.y = p[1];
                        if (x + 2 < X) t.z = p[2];
                        if (x + 3 < X) t.w = p[3];
                    }
                }
                v[r] = t;
            }
        }
    }

    __device__ inline void store(const f32x4 (&v)[NV], float (*S)[LDT], int tid) const {
        if (KCONTIG) {
            const int k4 = (tid % KL) * 4;
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int x = tid / KL + XR * r;
                S[k4 + 0][x] = v[r].x;
                S[k4 + 1][x] = v[r].y;
                S[k4 + 2][x] = v[r].z;
                S[k4 + 3][x] = v[r].w;
            }
        } else {
            const int x4 = (tid % XL) * 4;
#pragma unroll
            for (int r = 0; r < NV; ++r)
                *reinterpret_cast<f32x4*>(&S[tid / XL + KR * r][x4]) = v[r];
        }
    }
};

// acc += A(m0.., k) B(k, n0..) over k in [kbeg, kend) for one (64*MI) x 128 output tile.  Pipeline: global
// loads run TWO k-tiles ahead of the MFMAs (registers), LDS is double buffered, one barrier per k-tile.
// csum (TN form, optional): running column sums of the staged A tile (bias gradient).
// MODE 0: guarded loads (any shape / alignment; takes the unguarded path per tile when it can);  1: every k-tile full,
// both operand tiles interior and aligned: no guards, no divergent paths in the k loop;  2: as 1 for EDGE tiles (vectors
// outside the matrix predicated off).
template <bool AK, bool BK, int MI, int MODE>
__device__ inline void gemm_pipeline(f32x16 (&acc)[MI][2], float& csum, const bool want_csum,
                                     const TileLoader<AK, 64 * MI>& la, const TileLoader<BK, 128>& lb,
                                     const float* __restrict__ A, const RowMap& ra, const float* __restrict__ B, const RowMap& rb,
                                     int M, int N, int m0, int n0, int kbeg, int kend, int vecA, int vecB,
                                     float (*As)[BKT][LDT], float (*Bs)[BKT][LDT]) {
    constexpr int WM = 64 * MI;
    using LA = TileLoader<AK, WM>;
    using LB = TileLoader<BK, 128>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // MODE 1 / 2: every k-tile in [kbeg, kend) is full and both operand tiles are aligned: the k loop has
    // no guards and no divergent paths, so the compiler's waitcnt placement keeps two k-tiles of loads in flight
    // (a loop that also contains the guarded path gets conservative vmcnt(0) waits at every join)
    const int wm = (wave >> 1) * (32 * MI), wn = (wave & 1) * 64;
    const int li = lane & 31, lk = lane >> 5;
    const int nkt = (kend - kbeg + BKT - 1) / BKT;

    f32x4 ra0[LA::NV], rb0[LB::NV], ra1[LA::NV], rb1[LB::NV];      // tiles kt+1 and kt+2 in flight
    if (nkt > 0) {
        if (MODE) { la.template load_fast<MODE == 2>(ra0, kbeg); lb.template load_fast<MODE == 2>(rb0, kbeg); }
        else { la.load(ra0, A, ra, m0, M, kbeg, kend, tid, vecA); lb.load(rb0, B, rb, n0, N, kbeg, kend, tid, vecB); }
        if (nkt > 1) {
            if (MODE) { la.template load_fast<MODE == 2>(ra1, kbeg + BKT); lb.template load_fast<MODE == 2>(rb1, kbeg + BKT); }
            else { la.load(ra1, A, ra, m0, M, kbeg + BKT, kend, tid, vecA); lb.load(rb1, B, rb, n0, N, kbeg + BKT, kend, tid, vecB); }
        }
        la.store(ra0, As[0], tid);
        lb.store(rb0, Bs[0], tid);
    }
    __syncthreads();

#ifdef XPS_GSTAMP
    unsigned long long g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, s_ld = 0, s_mm = 0, s_st = 0, s_bar = 0;
#endif
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        GSTAMP(g0)
        if (kt + 2 < nkt) {
            if (MODE) { la.template load_fast<MODE == 2>(ra0, kbeg + (kt + 2) * BKT); lb.template load_fast<MODE == 2>(rb0, kbeg + (kt + 2) * BKT); }
            else { la.load(ra0, A, ra, m0, M, kbeg + (kt + 2) * BKT, kend, tid, vecA); lb.load(rb0, B, rb, n0, N, kbeg + (kt + 2) * BKT, kend, tid, vecB); }
        }
        GSTAMP(g1)
#pragma unroll
        for (int kk = 0; kk < BKT; kk += 2) {
            float a[MI], b[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = As[buf][kk + lk][wm + i * 32 + li];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[buf][kk + lk][wn + j * 32 + li];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (want_csum) {
            constexpr int G = 256 / WM, KG = BKT / G;          // k groups and rows per group
            const int cm = tid % WM, kh = (tid / WM) * KG;
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) csum += As[buf][kh + kk][cm];
        }
        GSTAMP(g2)
        if (kt + 1 < nkt) {
            la.store(ra1, As[buf ^ 1], tid);
            lb.store(rb1, Bs[buf ^ 1], tid);
#pragma unroll
            for (int r = 0; r < LA::NV; ++r) ra1[r] = ra0[r];
#pragma unroll
            for (int r = 0; r < LB::NV; ++r) rb1[r] = rb0[r];
        }
        GSTAMP(g3)
        __syncthreads();
#ifdef XPS_GSTAMP
        GSTAMP(g4)
        s_ld += g1 - g0; s_mm += g2 - g1; s_st += g3 - g2; s_bar += g4 - g3;
#endif
    }
#ifdef XPS_GSTAMP
    if (lane == 0) {
        const int wid = (((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave)) & 8191;
        g_gstamp[wid * 4 + 0] = s_ld; g_gstamp[wid * 4 + 1] = s_mm; g_gstamp[wid * 4 + 2] = s_st; g_gstamp[wid * 4 + 3] = s_bar;
    }
#endif
}

// acc += A(m0.., k) B(k, n0..) over k in [kbeg, kend) for one (64*MI) x 128 output tile.  Interior, aligned
// tiles run the unguarded pipeline over all full k-tiles and the guarded one only over a ragged k tail.
// EDGE: also instantiate the predicated pipeline for partial tiles (kernels launched for problems whose M / N are
// not tile multiples); the interior-only instantiation stays as compact as it was (it lost 3-9 % with the extra code).
template <bool AK, bool BK, int MI, bool EDGE = false>
__device__ inline void gemm_accumulate(f32x16 (&acc)[MI][2], float& csum, const bool want_csum,
                                       const float* __restrict__ A, const RowMap& ra, const float* __restrict__ B, const RowMap& rb,
                                       int M, int N, int K, int m0, int n0, int kbeg, int kend, int vecA, int vecB,
                                       float (*As)[BKT][LDT], float (*Bs)[BKT][LDT]) {
    TileLoader<AK, 64 * MI> la;
    TileLoader<BK, 128> lb;
    la.init(A, ra, m0, M, K, threadIdx.x, vecA);
    lb.init(B, rb, n0, N, K, threadIdx.x, vecB);
    const int kfull = kbeg + ((kend - kbeg) / BKT) * BKT;
    // Interior tiles: the k-contiguous (NT) form runs the branch-free pipeline over the full k-tiles and the guarded one
    // over a ragged tail (its guarded path is long); the [k][x] forms are faster with the single combined loop.
    // EDGE tiles (e.g. the 100-wide filter dimension in a 128-wide tile) used to fall to the guarded loop for the whole
    // k range: they now run the predicated branch-free pipeline (+10 % on the F = 100 shapes of the model).
    const bool both_fast = la.fast && lb.fast && kfull > kbeg;
    const bool both_full = la.full && lb.full;
    if (both_fast && both_full && AK && BK) {
        gemm_pipeline<AK, BK, MI, 1>(acc, csum, want_csum, la, lb, A, ra, B, rb, M, N, m0, n0, kbeg, kfull, vecA, vecB, As, Bs);
        if (kfull < kend)
            gemm_pipeline<AK, BK, MI, 0>(acc, csum, want_csum, la, lb, A, ra, B, rb, M, N, m0, n0, kfull, kend, vecA, vecB, As, Bs);
    } else if (EDGE && both_fast && !both_full) {
        gemm_pipeline<AK, BK, MI, 2>(acc, csum, want_csum, la, lb, A, ra, B, rb, M, N, m0, n0, kbeg, kfull, vecA, vecB, As, Bs);
        if (kfull < kend)
            gemm_pipeline<AK, BK, MI, 0>(acc, csum, want_csum, la, lb, A, ra, B, rb, M, N, m0, n0, kfull, kend, vecA, vecB, As, Bs);
    } else {
        gemm_pipeline<AK, BK, MI, 0>(acc, csum, want_csum, la, lb, A, ra, B, rb, M, N, m0, n0, kbeg, kend, vecA, vecB, As, Bs);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// bf16 split-product pipeline.  Every fp32 operand element x is split while it is staged into LDS:
//   hi = bf16(x),  lo = bf16(x - hi)      (x = hi + lo up to 2^-17 |x|)
// and a product tile is accumulated in fp32 as  lo_a*hi_b + hi_a*lo_b + hi_a*hi_b  on v_mfma_f32_32x32x16_bf16:
// three bf16 MFMAs (3 x 1/16 of the fp32-MFMA time) for a product exact to ~2^-16 relative (the dropped lo*lo term),
// i.e. 256x tighter than a plain bf16 GEMM and within the 1e-4 parity bar of the model.  The C/D layout of the
// 32x32 bf16 MFMA equals the fp32 form's, so gemm_store and every epilogue are shared with the fp32 pipeline.
// LDS image per operand and buffer (BfTile).  [x][k] operands: hi[128][24] + lo[128][24] shorts, row = x (m or n),
// 16 k-contiguous bf16 + 16 B pad -> 48-byte rows: the fragment reads (ds_read_b128: lane -> row, k half) are
// bank-conflict free.  [k][x] operands are NOT transposed on the way in: rows of k (320 B), fragments gathered with the
// transposing ds_read_b64_tr_b16 (bf_frag).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BFROW = 24;                                // shorts per LDS row (BKT = 16 used + 8 pad)
static_assert(BKT == 16, "the bf16 pipeline maps one k-tile to one 32x32x16 MFMA step");
// [k][x] image: W + 32 shorts per k row (W = 128: 320 B, W = 64: 192 B): the four k rows of a transposed read start 16 banks apart
template <int W>                                         // W = rows (x extent) of the staged operand tile: 64 or 128
struct BfTile {
    union {
        struct {                                         // [x][k] operands: row = x, 16 k-contiguous bf16 (+ pad)
            __bf16 hi[W][BFROW];
            __bf16 lo[W][BFROW];
        };
        struct {                                         // [k][x] operands: stored as they are loaded, row = k
            __bf16 thi[BKT][W + 32];
            __bf16 tlo[BKT][W + 32];
        };
    };
};
template <int MI>                                        // what a kernel declares in LDS: 2 buffers x (A: 64 MI rows, B: 128)
struct BfStage {
    BfTile<64 * MI> a[2];
    BfTile<128> b[2];
};

__device__ inline void bf_split(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

// Staging of one 16-byte vector.  fp32 operand: split its four values.  XPS_FMT_SPLIT4 operand (xps.h: the producer already
// wrote hi[0..3] | lo[0..3] into the 16 bytes): no arithmetic at all.  `pre` is block-uniform.
__device__ inline void stage_split(const f32x4& v, const bool pre, bf16x4& h, bf16x4& l) {
    if (pre) {
        // (element-wise shuffles: a bit_cast through a 2 x u64 vector returned the LOW half for both elements with this compiler)
        const bf16x8 q = __builtin_bit_cast(bf16x8, v);
        h = __builtin_shufflevector(q, q, 0, 1, 2, 3);
        l = __builtin_shufflevector(q, q, 4, 5, 6, 7);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { __bf16 a, b; bf_split(v[j], a, b); h[j] = a; l[j] = b; }
    }
}
// fp32 values of a staged vector (column sums of a split4 operand: hi + lo, within 2^-17 relative of the fp32 element)
__device__ inline f32x4 stage_values(const f32x4& v, const bool pre) {
    if (!pre) return v;
    bf16x4 h, l;
    stage_split(v, true, h, l);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (float)h[j] + (float)l[j];
    return o;
}
template <bool KCONTIG, int W>
__device__ inline void bf_store(const f32x4 (&v)[TileLoader<KCONTIG, W, true>::NV], BfTile<W>& S, int tid, const bool pre) {
    using L = TileLoader<KCONTIG, W, true>;
    if (KCONTIG) {
        const int k4 = (tid % KL) * 4;
#pragma unroll
        for (int r = 0; r < L::NV; ++r) {
            const int x = tid / KL + L::XR * r;
            bf16x4 h, l;
            stage_split(v[r], pre, h, l);
            *reinterpret_cast<bf16x4*>(&S.hi[x][k4]) = h;
            *reinterpret_cast<bf16x4*>(&S.lo[x][k4]) = l;
        }
    } else {
        // [k][x] operand: no transposition here — 8-byte stores of 4 x-consecutive bf16 into the k-row image (XL adjacent
        // lanes fill one row: conflict-free); the MFMA fragments are gathered by the transposing LDS read (bf_frag)
        const int x4 = (tid % L::XL) * 4;
#pragma unroll
        for (int r = 0; r < L::NV; ++r) {
            const int k = tid / L::XL + L::KR * r;
            bf16x4 h, l;
            stage_split(v[r], pre, h, l);
            *reinterpret_cast<bf16x4*>(&S.thi[k][x4]) = h;
            *reinterpret_cast<bf16x4*>(&S.tlo[k][x4]) = l;
        }
    }
}

// MFMA operand fragment (32x32x16: lane -> row x0 + (lane & 31), k = 8 (lane >> 5) + 0..7) of one staged tile.
// [x][k] image: one ds_read_b128.  [k][x] image: two ds_read_b64_tr_b16 — per 16-lane group the hardware reads a block of
// 4 k rows x 16 x columns and hands lane i column i (4 consecutive k of one x); lane 4q + p supplies the address of
// row q, columns 4p .. 4p + 3.  Needs EXEC = all ones (no divergence around the k loop).
template <bool KCONTIG, int W>
__device__ inline void bf_frag(const BfTile<W>& S, int x0, int lane, bf16x8& fh, bf16x8& fl) {
    if constexpr (KCONTIG) {
        fh = *reinterpret_cast<const bf16x8*>(&S.hi[x0 + (lane & 31)][(lane >> 5) * 8]);
        fl = *reinterpret_cast<const bf16x8*>(&S.lo[x0 + (lane & 31)][(lane >> 5) * 8]);
    } else {
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
        const int k0 = 8 * (g >> 1) + q, c0 = x0 + (g & 1) * 16 + 4 * pp;
        const bf16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(&S.thi[k0][c0]));
        const bf16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(&S.thi[k0 + 4][c0]));
        const bf16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(&S.tlo[k0][c0]));
        const bf16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(&S.tlo[k0 + 4][c0]));
        fh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        fl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

// Column sums of the A operand (bias gradient of the TN form) from the staging REGISTERS, before the split: every k-tile
// passes through them exactly once.  [k][x] form only: the thread's running sums are those of its 4 x columns over its
// NV k rows; the 256 / XL threads that share the x group are folded by the kernel (lane shuffle + LDS).
template <bool KCONTIG, int W>
__device__ inline void bf_colsum(const f32x4 (&v)[TileLoader<KCONTIG, W, true>::NV], f32x4& cs, const bool pre) {
    if constexpr (!KCONTIG) {           // (never requested for [x][k] operands)
#pragma unroll
        for (int r = 0; r < TileLoader<KCONTIG, W, true>::NV; ++r) cs += stage_values(v[r], pre);
    }
}

// same contract as gemm_pipeline (MODE 0 guarded / 1 interior / 2 predicated edge); csum: see bf_colsum
// ANYPRE = false: both operands are fp32 -- the k loop carries no operand-format flags at all (a run-time flag inside the loop
// cost the fp32-operand path 2.4 % of the cfg-2 step and 4 % of its weight-gradient launch: the branches keep the scheduler
// from interleaving staging and MFMAs; two loop copies inside one kernel raised its registers from 156 to 224).  ANYPRE = true
// (a kernel instantiation of its own, launched when an operand is XPS_FMT_SPLIT4): formats from bit 1 of the vec flags.
template <bool AK, bool BK, int MI, int MODE, bool ANYPRE = false>
__device__ inline void gemm_pipeline_bf(f32x16 (&acc)[MI][2], f32x4& csum, const bool want_csum,
                                        const TileLoader<AK, 64 * MI, true>& la, const TileLoader<BK, 128, true>& lb,
                                        const float* __restrict__ A, const RowMap& ra, const float* __restrict__ B, const RowMap& rb,
                                        int M, int N, int m0, int n0, int kbeg, int kend, int vecA, int vecB, BfStage<MI>& S) {
    constexpr int WM = 64 * MI;
    using LA = TileLoader<AK, WM, true>;
    using LB = TileLoader<BK, 128, true>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * (32 * MI), wn = (wave & 1) * 64;
    const int li = lane & 31, lk = lane >> 5;
    const int nkt = (kend - kbeg + BKT - 1) / BKT;
    const bool preA = ANYPRE && (vecA & 2) != 0, preB = ANYPRE && (vecB & 2) != 0;

    f32x4 ra0[LA::NV], rb0[LB::NV], ra1[LA::NV], rb1[LB::NV];      // tiles kt+1 and kt+2 in flight
    if (nkt > 0) {
        if (MODE) { la.template load_fast<MODE == 2>(ra0, kbeg); lb.template load_fast<MODE == 2>(rb0, kbeg); }
        else { la.load(ra0, A, ra, m0, M, kbeg, kend, tid, vecA); lb.load(rb0, B, rb, n0, N, kbeg, kend, tid, vecB); }
        if (nkt > 1) {
            if (MODE) { la.template load_fast<MODE == 2>(ra1, kbeg + BKT); lb.template load_fast<MODE == 2>(rb1, kbeg + BKT); }
            else { la.load(ra1, A, ra, m0, M, kbeg + BKT, kend, tid, vecA); lb.load(rb1, B, rb, n0, N, kbeg + BKT, kend, tid, vecB); }
        }
        bf_store<AK, WM>(ra0, S.a[0], tid, preA);
        bf_store<BK, 128>(rb0, S.b[0], tid, preB);
        if (want_csum) bf_colsum<AK, WM>(ra0, csum, preA);
    }
    __syncthreads();
#ifdef XPS_GSTAMP
    unsigned long long g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, s_ld = 0, s_mm = 0, s_st = 0, s_bar = 0;
#endif
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        GSTAMP(g0)
        if (kt + 2 < nkt) {
            if (MODE) { la.template load_fast<MODE == 2>(ra0, kbeg + (kt + 2) * BKT); lb.template load_fast<MODE == 2>(rb0, kbeg + (kt + 2) * BKT); }
            else { la.load(ra0, A, ra, m0, M, kbeg + (kt + 2) * BKT, kend, tid, vecA); lb.load(rb0, B, rb, n0, N, kbeg + (kt + 2) * BKT, kend, tid, vecB); }
        }
        GSTAMP(g1)
        {
            bf16x8 ah[MI], al[MI], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < MI; ++i) bf_frag<AK, 64 * MI>(S.a[buf], wm + i * 32, lane, ah[i], al[i]);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf_frag<BK, 128>(S.b[buf], wn + j * 32, lane, bh[j], bl[j]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        GSTAMP(g2)
        if (kt + 1 < nkt) {
            bf_store<AK, WM>(ra1, S.a[buf ^ 1], tid, preA);
            bf_store<BK, 128>(rb1, S.b[buf ^ 1], tid, preB);
            if (want_csum) bf_colsum<AK, WM>(ra1, csum, preA);
#pragma unroll
            for (int r = 0; r < LA::NV; ++r) ra1[r] = ra0[r];
#pragma unroll
            for (int r = 0; r < LB::NV; ++r) rb1[r] = rb0[r];
        }
        GSTAMP(g3)
        __syncthreads();
#ifdef XPS_GSTAMP
        GSTAMP(g4)
        s_ld += g1 - g0; s_mm += g2 - g1; s_st += g3 - g2; s_bar += g4 - g3;
#endif
    }
#ifdef XPS_GSTAMP
    if (lane == 0) {
        const int wid = (((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave)) & 8191;
        g_gstamp[wid * 4 + 0] = s_ld; g_gstamp[wid * 4 + 1] = s_mm; g_gstamp[wid * 4 + 2] = s_st; g_gstamp[wid * 4 + 3] = s_bar;
    }
#endif
}

template <bool AK, bool BK, int MI, bool EDGE = false, bool ANYPRE = false>
__device__ inline void gemm_accumulate_bf(f32x16 (&acc)[MI][2], f32x4& csum, const bool want_csum,
                                          const float* __restrict__ A, const RowMap& ra, const float* __restrict__ B, const RowMap& rb,
                                          int M, int N, int K, int m0, int n0, int kbeg, int kend, int vecA, int vecB, BfStage<MI>& S) {
    TileLoader<AK, 64 * MI, true> la;
    TileLoader<BK, 128, true> lb;
    la.init(A, ra, m0, M, K, threadIdx.x, vecA);
    lb.init(B, rb, n0, N, K, threadIdx.x, vecB);
    const int kfull = kbeg + ((kend - kbeg) / BKT) * BKT;
    const bool both_fast = la.fast && lb.fast && kfull > kbeg;
    const bool both_full = la.full && lb.full;
    if (both_fast && both_full) {
        gemm_pipeline_bf<AK, BK, MI, 1, ANYPRE>(acc, csum, want_csum, la, lb, A, ra, B, rb, M, N, m0, n0, kbeg, kfull, vecA, vecB, S);
        if (kfull < kend)
            gemm_pipeline_bf<AK, BK, MI, 0, ANYPRE>(acc, csum, want_csum, la, lb, A, ra, B, rb, M, N, m0, n0, kfull, kend, vecA, vecB, S);
    } else if (EDGE && both_fast && !both_full) {
        gemm_pipeline_bf<AK, BK, MI, 2, ANYPRE>(acc, csum, want_csum, la, lb, A, ra, B, rb, M, N, m0, n0, kbeg, kfull, vecA, vecB, S);
        if (kfull < kend)
            gemm_pipeline_bf<AK, BK, MI, 0, ANYPRE>(acc, csum, want_csum, la, lb, A, ra, B, rb, M, N, m0, n0, kfull, kend, vecA, vecB, S);
    } else {
        gemm_pipeline_bf<AK, BK, MI, 0, ANYPRE>(acc, csum, want_csum, la, lb, A, ra, B, rb, M, N, m0, n0, kbeg, kend, vecA, vecB, S);
    }
}

// the staging memory of a tile kernel in either precision mode (BF = false: the fp32 k-major tiles)
template <bool BF, int MI = 2> struct TileMem;
template <int MI> struct TileMem<false, MI> {
    float As[2][BKT][LDT];
    float Bs[2][BKT][LDT];
};
template <int MI> struct TileMem<true, MI> {
    BfStage<MI> st;
};
template <bool AK, bool BK, int MI, bool EDGE, bool BF, bool ANYPRE = false>
__device__ inline void gemm_accumulate_any(f32x16 (&acc)[MI][2], float& csum, f32x4& csum4, const bool want_csum,
                                           const float* __restrict__ A, const RowMap& ra, const float* __restrict__ B, const RowMap& rb,
                                           int M, int N, int K, int m0, int n0, int kbeg, int kend, int vecA, int vecB,
                                           TileMem<BF, MI>& mem) {
    // csum (fp32 pipeline: one column, half the k rows per thread) / csum4 (bf16 pipeline: see bf_colsum)
    if constexpr (BF) gemm_accumulate_bf<AK, BK, MI, EDGE, ANYPRE>(acc, csum4, want_csum, A, ra, B, rb, M, N, K, m0, n0, kbeg, kend, vecA, vecB, mem.st);
    else gemm_accumulate<AK, BK, MI, EDGE>(acc, csum, want_csum, A, ra, B, rb, M, N, K, m0, n0, kbeg, kend, vecA, vecB, mem.As, mem.Bs);
}

// C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
template <int MI>
__device__ inline void gemm_store(const f32x16 (&acc)[MI][2], float* __restrict__ C, const RowMap& rc,
                                  const float* __restrict__ bias, int M, int N, int m0, int n0, int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * (32 * MI), wn = (wave & 1) * 64;
    const int li = lane & 31, lk = lane >> 5;
    // Interior tile of a plainly strided C (one row group): no bounds checks and no row-map divisions.  The general path
    // below costs ~55 instructions per stored element (two integer divisions per row); that was 20 % of a tile's time with
    // the fp32 k loop and more than half of it with the bf16 one.
    if (rc.rpg >= M && m0 + 64 * MI <= M && n0 + 128 <= N) {
        float* cbase = C + (long long)(m0 + wm + 4 * lk) * rc.ld + n0 + wn + li;
        float bv[2] = {0.f, 0.f};
        if (bias) { bv[0] = bias[n0 + wn + li]; bv[1] = bias[n0 + wn + 32 + li]; }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* crow = cbase + (long long)(i * 32 + (r & 3) + 8 * (r >> 2)) * rc.ld;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v = acc[i][j][r] + bv[j];
                    if (accumulate) v += crow[j * 32];
                    __builtin_nontemporal_store(v, &crow[j * 32]);
                }
            }
        return;
    }
    // edge tiles / mapped rows: bounds checks; the row-map divisions only when C really has row groups
    const bool plain = rc.rpg >= M;
    const int c0 = n0 + wn + li, c1 = c0 + 32;
    float bv[2] = {0.f, 0.f};
    if (bias) { bv[0] = c0 < N ? bias[c0] : 0.f; bv[1] = c1 < N ? bias[c1] : 0.f; }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row >= M) continue;
            float* crow = C + (plain ? (long long)row * rc.ld : rc.off(row));
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = j ? c1 : c0;
                if (col < N) {
                    float v = acc[i][j][r] + bv[j];
                    if (accumulate) v += crow[col];
                    __builtin_nontemporal_store(v, &crow[col]);        // streamed once: -1 % on the step vs a plain store
                }
            }
        }
    }
}

template <int MI>
__device__ inline void zero_acc(f32x16 (&acc)[MI][2]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}


}  // namespace xps_tile
