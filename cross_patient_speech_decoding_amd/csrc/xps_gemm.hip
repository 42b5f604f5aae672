// Dense fp32 GEMMs on the f32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32 fma
// chains, 64 FLOP/clk/SIMD = the fp32 matrix peak of gfx950 (157 TFLOP/s).
//
// One LDS-tiled kernel, 128 x 128 x 16 block tile, 4 waves each owning a 64 x 64
// sub-tile (2 x 2 MFMA 32x32 tiles).  Both operands are staged k-major in LDS
// ([k][m] / [k][n], row stride 132 floats) so that the MFMA operand reads are
// conflict-free ds_read_b32 (lanes 0-31 consecutive floats; lanes 32-63 the next k
// row).  Global loads are 16-byte vectors along whichever index is contiguous in
// memory; the next tile is fetched into registers while the current one is
// multiplied (one barrier per k-tile, two LDS buffers).
//
// Operand forms (template flags):  AK = A is contiguous along k ([m][k]),
// otherwise [k][m];  BK likewise for B.  NT = <1,1>, NN = <1,0>, TN = <0,0>.
// Rows are addressed through a two-level row map (see xps.h) so the same kernel
// reads convolution windows of a (trial, time, channel) tensor and writes
// time-major outputs.
#include <stdlib.h>
#include "xps_common.h"

namespace {

#ifndef XPS_BKT
#define XPS_BKT 16
#endif
#ifndef XPS_GEMM_WAVES
#define XPS_GEMM_WAVES 1
#endif
constexpr int BM = 128, BN = 128, BKT = XPS_BKT, LDT = 132;

constexpr int NV = BKT / 8;          // 16-byte vectors per thread, operand and k-tile
constexpr int KL = BKT / 4;          // KCONTIG: lanes covering one row's k range
constexpr int XR = 256 / KL;         // KCONTIG: rows of x covered per pass

template <bool KCONTIG>
struct TileLoader {
    // KCONTIG : matrix stored [x][k]  (x = m or n), 16-byte vectors along k
    //           thread -> (x = tid / KL (+ XR per pass), k4 = (tid % KL) * 4)
    // !KCONTIG: matrix stored [k][x], 16-byte vectors along x
    //           thread -> (k = tid >> 5 (+ 8 per pass), x4 = (tid & 31) * 4)
    // `fast` (block-uniform): the 128-wide x range is fully inside the matrix, vectors are aligned and
    // the row map is a plain leading dimension -> no per-element guards, no divisions in the k loop.
    const float* base[NV];  // fast path: per-thread base pointers (k = 0)
    long long xoff[NV];     // guarded KCONTIG path: row offset of x (or -1)
    long long kstride;      // fast !KCONTIG: elements between consecutive k rows
    bool fast;

    __device__ inline void init(const float* __restrict__ P, const RowMap& rm, int x0, int X, int rows_k, int tid, bool vec) {
        if (KCONTIG) {
            fast = vec && (x0 + 128 <= X);
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int x = x0 + tid / KL + XR * r;
                xoff[r] = (x < X) ? rm.off(x) : -1;
                base[r] = P + (xoff[r] >= 0 ? xoff[r] : 0) + (tid % KL) * 4;
            }
            kstride = 1;
        } else {
            fast = vec && (x0 + 128 <= X) && (rm.rpg >= rows_k);
            kstride = rm.ld;
#pragma unroll
            for (int r = 0; r < NV; ++r)
                base[r] = P + (long long)((tid >> 5) + 8 * r) * rm.ld + x0 + (tid & 31) * 4;
        }
    }

    __device__ inline void load(float4 (&v)[NV], const float* __restrict__ P, const RowMap& rm, int x0, int X,
                                int kt0, int kend, int tid, bool vec) const {
        if (fast && kt0 + BKT <= kend) {
#pragma unroll
            for (int r = 0; r < NV; ++r)
                v[r] = *reinterpret_cast<const float4*>(base[r] + (KCONTIG ? (long long)kt0 : (long long)kt0 * kstride));
            return;
        }
        if (KCONTIG) {
            const int k = kt0 + (tid % KL) * 4;
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (xoff[r] >= 0) {
                    const float* p = P + xoff[r] + k;
                    if (vec && k + 3 < kend) {
                        t = *reinterpret_cast<const float4*>(p);
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
            const int x = x0 + (tid & 31) * 4;
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int k = kt0 + (tid >> 5) + 8 * r;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < kend && x < X) {
                    const float* p = P + rm.off(k) + x;
                    if (vec && x + 3 < X) {
                        t = *reinterpret_cast<const float4*>(p);
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

    __device__ inline void store(const float4 (&v)[NV], float (*S)[LDT], int tid) const {
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
            const int x4 = (tid & 31) * 4;
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int k = (tid >> 5) + 8 * r;
                *reinterpret_cast<float4*>(&S[k][x4]) = v[r];
            }
        }
    }
};

// One 128 x 128 output tile over k in [kbeg, kend).  Pipeline: global loads run TWO k-tiles ahead of
// the MFMAs (registers), LDS is double buffered, one barrier per k-tile.
template <bool AK, bool BK>
__device__ inline void gemm_tile(const float* __restrict__ A, const RowMap& ra, const float* __restrict__ B, const RowMap& rb,
                                 float* __restrict__ C, const RowMap& rc, const float* __restrict__ bias,
                                 int M, int N, int K, int m0, int n0, int kbeg, int kend, int accumulate,
                                 int vecA, int vecB, float* __restrict__ colsum_out,
                                 float (*As)[BKT][LDT], float (*Bs)[BKT][LDT]) {
    // colsum_out (or null): colsum_out[m - m0 ...] receives sum_k A(k, m) over this block's k range,
    // folded from the LDS copy of the A tile (thread -> column tid & 127, k half tid >> 7)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float csum = 0.f;

    TileLoader<AK> la;
    TileLoader<BK> lb;
    la.init(A, ra, m0, M, K, tid, vecA);
    lb.init(B, rb, n0, N, K, tid, vecB);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int li = lane & 31, lk = lane >> 5;
    const int nkt = (kend - kbeg + BKT - 1) / BKT;

    float4 ra0[NV], rb0[NV], ra1[NV], rb1[NV];      // tiles kt+1 and kt+2 in flight
    if (nkt > 0) {
        la.load(ra0, A, ra, m0, M, kbeg, kend, tid, vecA);
        lb.load(rb0, B, rb, n0, N, kbeg, kend, tid, vecB);
        if (nkt > 1) {
            la.load(ra1, A, ra, m0, M, kbeg + BKT, kend, tid, vecA);
            lb.load(rb1, B, rb, n0, N, kbeg + BKT, kend, tid, vecB);
        }
        la.store(ra0, As[0], tid);
        lb.store(rb0, Bs[0], tid);
    }
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        // registers: (ra1, rb1) hold tile kt+1; request tile kt+2 into (ra0, rb0)
        if (kt + 2 < nkt) {
            la.load(ra0, A, ra, m0, M, kbeg + (kt + 2) * BKT, kend, tid, vecA);
            lb.load(rb0, B, rb, n0, N, kbeg + (kt + 2) * BKT, kend, tid, vecB);
        }
#pragma unroll
        for (int kk = 0; kk < BKT; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[buf][kk + lk][wm + i * 32 + li];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[buf][kk + lk][wn + j * 32 + li];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (colsum_out) {
            const int cm = tid & 127, kh = (tid >> 7) * (BKT / 2);
#pragma unroll
            for (int kk = 0; kk < BKT / 2; ++kk) csum += As[buf][kh + kk][cm];
        }
        if (kt + 1 < nkt) {
            la.store(ra1, As[buf ^ 1], tid);
            lb.store(rb1, Bs[buf ^ 1], tid);
#pragma unroll
            for (int r = 0; r < NV; ++r) { ra1[r] = ra0[r]; rb1[r] = rb0[r]; }
        }
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (row >= M) continue;
            float* crow = C + rc.off(row);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn + j * 32 + li;
                if (col < N) {
                    float v = acc[i][j][r];
                    if (bias) v += bias[col];
                    if (accumulate) v += crow[col];
                    crow[col] = v;
                }
            }
        }
    }
    if (colsum_out) {
        __syncthreads();
        float* red = &Bs[0][0][0];
        red[tid] = csum;
        __syncthreads();
        if (tid < 128 && m0 + tid < M) colsum_out[tid] = red[tid] + red[tid + 128];
    }
}

template <bool AK, bool BK>
__global__ __launch_bounds__(256, XPS_GEMM_WAVES) void gemm_f32_kernel(
    const float* __restrict__ A, RowMap ra, const float* __restrict__ B, RowMap rb,
    float* __restrict__ C, RowMap rc, const float* __restrict__ bias,
    int M, int N, int K, int kchunk, long long slab_stride, int accumulate, int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) float As[2][BKT][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2][BKT][LDT];
    const int kbeg = blockIdx.z * kchunk;
    const int kend = min(K, kbeg + kchunk);
    gemm_tile<AK, BK>(A, ra, B, rb, C + (long long)blockIdx.z * slab_stride, rc, bias, M, N, K,
                      blockIdx.y * BM, blockIdx.x * BN, kbeg, kend, accumulate, vecA, vecB, nullptr, As, Bs);
}

// Grouped TN GEMM: up to TN_MAXP weight-gradient problems  C_p = A_p^T B_p  (+ column sums of A_p
// from the LDS copy of the A tiles = the bias gradient) in ONE launch; every block owns one
// (problem, tile, k-split) and writes a partial slab.
constexpr int TN_MAXP = 12;
struct TnProb {
    const float* A; const float* B; float* C; float* colsum;
    RowMap ra, rb, rc;
    int M, N, K, accumulate;
    int splits, kchunk, tiles_n, vecA, vecB;
    long long slab_off;       // float offset of this problem's slabs in the workspace
    int block_start;          // first flat block id
    int pad_;
};
struct TnGroup {
    TnProb p[TN_MAXP];
    int n;
    int total_blocks;
    long long total_out;      // sum of M * (N + has_colsum)
};

__global__ __launch_bounds__(256, XPS_GEMM_WAVES) void gemm_tn_grouped_kernel(TnGroup g, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) float As[2][BKT][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2][BKT][LDT];
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < g.n; ++i)
        if ((int)blockIdx.x >= g.p[i].block_start) pi = i;
    const TnProb& P = g.p[pi];
    const int local = blockIdx.x - P.block_start;
    const int Nout = P.N + (P.colsum ? 1 : 0);          // slab row = N products + 1 column sum
    const int tiles_m = (P.M + BM - 1) / BM;
    const int tile = local % (tiles_m * P.tiles_n), z = local / (tiles_m * P.tiles_n);
    const int tm = tile / P.tiles_n, tn = tile % P.tiles_n;
    const int kbeg = z * P.kchunk;
    const int kend = min(P.K, kbeg + P.kchunk);
    RowMap rs;
    rs.gs = 0; rs.ld = P.N; rs.rpg = 1 << 30;
    float* slab = ws + P.slab_off + (long long)z * P.M * Nout;
    float* cs = (P.colsum && tn == 0) ? slab + (long long)P.M * P.N + tm * BM : nullptr;
    gemm_tile<false, false>(P.A, P.ra, P.B, P.rb, slab, rs, nullptr, P.M, P.N, P.K, tm * BM, tn * BN, kbeg, kend, 0,
                            P.vecA, P.vecB, cs, As, Bs);
}

__global__ void gemm_tn_grouped_reduce(TnGroup g, const float* __restrict__ ws) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.total_out) return;
    int pi = 0;
    long long base = 0;
#pragma unroll 1
    for (int i = 0; i < g.n; ++i) {
        const long long sz = (long long)g.p[i].M * (g.p[i].N + (g.p[i].colsum ? 1 : 0));
        if (idx < base + sz) { pi = i; break; }
        base += sz;
    }
    const TnProb& P = g.p[pi];
    const int Nout = P.N + (P.colsum ? 1 : 0);
    const long long e = idx - base;                       // [0, M*N): products, then M column sums
    const float* sl = ws + P.slab_off + e;
    const long long stride = (long long)P.M * Nout;
    float s = 0.f;
    for (int z = 0; z < P.splits; ++z) s += sl[z * stride];
    const long long mn = (long long)P.M * P.N;
    float* dst = (e < mn) ? (P.C + P.rc.off((int)(e / P.N)) + (int)(e % P.N)) : (P.colsum + (e - mn));
    if (P.accumulate) s += *dst;
    *dst = s;
}

__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, long long slab_stride,
                                     float* __restrict__ C, RowMap rc, int M, int N, int accumulate) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * N) return;
    int m = (int)(idx / N), n = (int)(idx % N);
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slabs[(long long)z * slab_stride + idx];
    float* p = C + rc.off(m) + n;
    if (accumulate) s += *p;
    *p = s;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool map_vec_ok(const float* p, const RowMap& r) {
    return aligned16(p) && (r.ld % 4 == 0) && (r.gs % 4 == 0);
}

int tn_splits(int M, int N, int K) {
    long long tiles = (long long)cdiv(M, BM) * cdiv(N, BN);
    int want = (int)((768 + tiles - 1) / tiles);
    int maxs = cdiv(K, 256);
    int s = want < maxs ? want : maxs;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return s;
}

}  // namespace

extern "C" int xps_gemm_nt_f32(const float* A, const xps_rowmap* ra_, const float* B, const xps_rowmap* rb_,
                               float* C, const xps_rowmap* rc_, const float* bias,
                               int M, int N, int K, int accumulate, void* stream) {
    XPS_CHECK_ARG(A && B && C && ra_ && rb_ && rc_, "null argument");
    XPS_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative size");
    if (M == 0 || N == 0) return XPS_OK;
    RowMap ra = to_rowmap(ra_), rb = to_rowmap(rb_), rc = to_rowmap(rc_);
    dim3 grid(cdiv(N, BN), cdiv(M, BM), 1);
    hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream,
                       A, ra, B, rb, C, rc, bias, M, N, K, ((K + BKT - 1) / BKT) * BKT + BKT, 0LL, accumulate,
                       (int)map_vec_ok(A, ra), (int)map_vec_ok(B, rb));
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_gemm_nn_f32(const float* A, const xps_rowmap* ra_, const float* B, const xps_rowmap* rb_,
                               float* C, const xps_rowmap* rc_,
                               int M, int N, int K, int accumulate, void* stream) {
    XPS_CHECK_ARG(A && B && C && ra_ && rb_ && rc_, "null argument");
    XPS_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative size");
    if (M == 0 || N == 0) return XPS_OK;
    RowMap ra = to_rowmap(ra_), rb = to_rowmap(rb_), rc = to_rowmap(rc_);
    dim3 grid(cdiv(N, BN), cdiv(M, BM), 1);
    hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream,
                       A, ra, B, rb, C, rc, (const float*)nullptr, M, N, K, ((K + BKT - 1) / BKT) * BKT + BKT, 0LL,
                       accumulate, (int)map_vec_ok(A, ra), (int)map_vec_ok(B, rb));
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" size_t xps_gemm_tn_f32_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 16;
    int s = tn_splits(M, N, K);
    return (size_t)s * (size_t)M * (size_t)N * sizeof(float) + 16;
}

extern "C" int xps_gemm_tn_f32(const float* A, const xps_rowmap* ra_, const float* B, const xps_rowmap* rb_,
                               float* C, const xps_rowmap* rc_,
                               int M, int N, int K, int accumulate,
                               void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(A && B && C && ra_ && rb_ && rc_, "null argument");
    XPS_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative size");
    if (M == 0 || N == 0) return XPS_OK;
    RowMap ra = to_rowmap(ra_), rb = to_rowmap(rb_), rc = to_rowmap(rc_);
    const int splits = (K > 0) ? tn_splits(M, N, K) : 1;
    if (workspace_bytes < xps_gemm_tn_f32_workspace(M, N, K) || !workspace) {
        xps_set_error("xps_gemm_tn_f32: workspace too small (%zu < %zu)", workspace_bytes,
                      xps_gemm_tn_f32_workspace(M, N, K));
        return XPS_E_WORKSPACE;
    }
    XPS_CHECK_ARG(aligned16(workspace), "workspace must be 16-byte aligned");
    int kchunk = ((cdiv(K > 0 ? K : 1, splits) + BKT - 1) / BKT) * BKT;
    float* slabs = reinterpret_cast<float*>(workspace);
    RowMap rs;
    rs.gs = 0; rs.ld = N; rs.rpg = 1 << 30;
    const long long slab_stride = (long long)M * N;
    dim3 grid(cdiv(N, BN), cdiv(M, BM), splits);
    hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream,
                       A, ra, B, rb, slabs, rs, (const float*)nullptr, M, N, K, kchunk, slab_stride, 0,
                       (int)map_vec_ok(A, ra), (int)map_vec_ok(B, rb));
    XPS_CHECK_LAUNCH();
    long long total = (long long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       slabs, splits, slab_stride, C, rc, M, N, accumulate);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

namespace {
int build_group(const xps_tn_problem* probs, int n, TnGroup& g, size_t& ws_floats) {
    if (n < 1 || n > TN_MAXP) return -1;
    static const int target_blocks = [] {
        const char* e = getenv("XPS_TN_BLOCKS");
        int v = e ? atoi(e) : 0;
        return v > 0 ? v : 768;
    }();
    long long tiles_total = 0;
    for (int i = 0; i < n; ++i) tiles_total += (long long)cdiv(probs[i].M, BM) * cdiv(probs[i].N, BN);
    g.n = n;
    int blocks = 0;
    long long off = 0, out = 0;
    for (int i = 0; i < n; ++i) {
        const xps_tn_problem& q = probs[i];
        if (!q.A || !q.B || !q.C || q.M < 1 || q.N < 1 || q.K < 0) return -1;
        TnProb& P = g.p[i];
        P.A = q.A; P.B = q.B; P.C = q.C; P.colsum = q.colsum_a;
        P.ra = to_rowmap(&q.ra); P.rb = to_rowmap(&q.rb); P.rc = to_rowmap(&q.rc);
        P.M = q.M; P.N = q.N; P.K = q.K; P.accumulate = q.accumulate;
        const int Nout = q.N + (q.colsum_a ? 1 : 0);
        const int tiles = cdiv(q.M, BM) * cdiv(q.N, BN);
        // share the block budget among the problems in proportion to their tiles; >= 64 rows per split
        int want = (int)((target_blocks * (long long)tiles / (tiles_total > 0 ? tiles_total : 1) + tiles - 1) / tiles);
        int maxs = cdiv(q.K > 0 ? q.K : 1, 64);
        int sp = want < maxs ? want : maxs;
        if (sp < 1) sp = 1;
        if (sp > 256) sp = 256;
        P.splits = sp;
        P.kchunk = ((cdiv(q.K > 0 ? q.K : 1, sp) + BKT - 1) / BKT) * BKT;
        P.tiles_n = cdiv(q.N, BN);
        P.vecA = (int)map_vec_ok(q.A, P.ra);
        P.vecB = (int)map_vec_ok(q.B, P.rb);
        P.slab_off = off;
        P.block_start = blocks;
        P.pad_ = 0;
        blocks += tiles * sp;
        off += (long long)sp * q.M * Nout;
        out += (long long)q.M * Nout;
    }
    g.total_blocks = blocks;
    g.total_out = out;
    ws_floats = (size_t)off;
    return 0;
}
}  // namespace

extern "C" size_t xps_gemm_tn_grouped_f32_workspace(const xps_tn_problem* probs, int n) {
    TnGroup g;
    size_t fl = 0;
    if (!probs || build_group(probs, n, g, fl)) return 0;
    return fl * sizeof(float) + 16;
}

extern "C" int xps_gemm_tn_grouped_f32(const xps_tn_problem* probs, int n, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    XPS_CHECK_ARG(probs, "null argument");
    TnGroup g;
    size_t fl = 0;
    if (build_group(probs, n, g, fl)) {
        xps_set_error("xps_gemm_tn_grouped_f32: invalid problem list (1..%d problems, non-null pointers, M,N >= 1)", TN_MAXP);
        return XPS_E_INVALID;
    }
    if (!workspace || workspace_bytes < fl * sizeof(float) + 16 || !aligned16(workspace)) {
        xps_set_error("xps_gemm_tn_grouped_f32: workspace too small or misaligned");
        return XPS_E_WORKSPACE;
    }
    hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(g.total_blocks), dim3(256), 0, (hipStream_t)stream, g, (float*)workspace);
    XPS_CHECK_LAUNCH();
    hipLaunchKernelGGL(gemm_tn_grouped_reduce, dim3(cdiv(g.total_out, 256)), dim3(256), 0, (hipStream_t)stream, g,
                       (const float*)workspace);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}
