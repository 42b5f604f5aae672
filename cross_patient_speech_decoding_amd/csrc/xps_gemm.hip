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
#include "xps_common.h"

namespace {

constexpr int BM = 128, BN = 128, BKT = 16, LDT = 132;

template <bool KCONTIG>
struct TileLoader {
    // KCONTIG : matrix stored [x][k]  (x = m or n), 16-byte vectors along k
    //           thread -> (x = tid>>2 (+64), k4 = (tid&3)*4)
    // !KCONTIG: matrix stored [k][x], 16-byte vectors along x
    //           thread -> (k = tid>>5 (+8), x4 = (tid&31)*4)
    float4 v[2];
    long long xoff[2];   // KCONTIG: row offset of x (fixed for the whole k loop)

    __device__ inline void init(const RowMap& rm, int x0, int X, int tid) {
        if (KCONTIG) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                int x = x0 + (tid >> 2) + 64 * r;
                xoff[r] = (x < X) ? rm.off(x) : -1;
            }
        }
    }

    __device__ inline void load(const float* __restrict__ P, const RowMap& rm, int x0, int X,
                                int kt0, int kend, int tid, bool vec) {
        if (KCONTIG) {
            const int k = kt0 + (tid & 3) * 4;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
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
            for (int r = 0; r < 2; ++r) {
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

    __device__ inline void store(float (*S)[LDT], int tid) const {
        if (KCONTIG) {
            const int k4 = (tid & 3) * 4;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int x = (tid >> 2) + 64 * r;
                S[k4 + 0][x] = v[r].x;
                S[k4 + 1][x] = v[r].y;
                S[k4 + 2][x] = v[r].z;
                S[k4 + 3][x] = v[r].w;
            }
        } else {
            const int x4 = (tid & 31) * 4;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int k = (tid >> 5) + 8 * r;
                *reinterpret_cast<float4*>(&S[k][x4]) = v[r];
            }
        }
    }
};

template <bool AK, bool BK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(
    const float* __restrict__ A, RowMap ra, const float* __restrict__ B, RowMap rb,
    float* __restrict__ C, RowMap rc, const float* __restrict__ bias,
    int M, int N, int K, int kchunk, long long slab_stride, int accumulate, int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) float As[2][BKT][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[2][BKT][LDT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * kchunk;
    const int kend = min(K, kbeg + kchunk);
    C += (long long)blockIdx.z * slab_stride;

    TileLoader<AK> la;
    TileLoader<BK> lb;
    la.init(ra, m0, M, tid);
    lb.init(rb, n0, N, tid);

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

    if (nkt > 0) {
        la.load(A, ra, m0, M, kbeg, kend, tid, vecA);
        lb.load(B, rb, n0, N, kbeg, kend, tid, vecB);
        la.store(As[0], tid);
        lb.store(Bs[0], tid);
    }
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) {
            la.load(A, ra, m0, M, kbeg + (kt + 1) * BKT, kend, tid, vecA);
            lb.load(B, rb, n0, N, kbeg + (kt + 1) * BKT, kend, tid, vecB);
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
        if (kt + 1 < nkt) {
            la.store(As[buf ^ 1], tid);
            lb.store(Bs[buf ^ 1], tid);
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
                       A, ra, B, rb, C, rc, bias, M, N, K, ((K + 15) / 16) * 16 + 16, 0LL, accumulate,
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
                       A, ra, B, rb, C, rc, (const float*)nullptr, M, N, K, ((K + 15) / 16) * 16 + 16, 0LL,
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
    int kchunk = ((cdiv(K > 0 ? K : 1, splits) + 15) / 16) * 16;
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
