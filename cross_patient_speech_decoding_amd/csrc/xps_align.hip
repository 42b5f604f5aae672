// Alignment kernels (CCA / MCCA / joint PCA / PCA) for gfx950.
//
//  k1  xps_cnd_avg_*      segmented per-condition mean of (trial x time x channel)
//                         tensors, coalesced along the (time*channel) index, sequential
//                         in trial order in the input dtype = np.mean bit for bit.
//                         HBM-bound: N*T*d*itemsize read + n_c*T*d*8 written.
//  k2  xps_xcov_f64       centred Gram / cross-covariance  (A-mean)^T (B-mean)  on the
//                         f64 MFMA (v_mfma_f64_16x16x4_f64), fp32 or fp64 inputs converted
//                         and centred while staged to LDS, deterministic split-K slabs.
//  k3/k5 xps_jacobi_*     one-sided Jacobi (Hestenes) rotations in fp64: SVD of small
//                         dense matrices and eigendecomposition of PSD matrices (the
//                         whitened generalised eigenproblem of MCCA, covariance of PCA).
//  k4  xps_apply_f64      batched transform apply  (X - mean) W  (same MFMA kernel).
//
// f64 MFMA operand mapping (16x16x4, lane l: n = l & 15, kq = l >> 4):
//   A[row n][k kq], B[k kq][col n], D[row = kq + 4*i][col = n] in register i.
#include "xps_common.h"

namespace {

// ------------------------------------------------------------------ k1 ----------
template <typename T>
__global__ void cnd_avg_kernel(const T* __restrict__ data, const int* __restrict__ order,
                               const int* __restrict__ start, double* __restrict__ out, long long row_len) {
    const int c = blockIdx.y;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= row_len) return;
    const int s0 = start[c], s1 = start[c + 1];
    if (s1 <= s0) { out[(long long)c * row_len + e] = 0.0; return; }
    T acc = data[(long long)order[s0] * row_len + e];
    for (int i = s0 + 1; i < s1; ++i) acc = acc + data[(long long)order[i] * row_len + e];
    const T mean = acc / (T)(s1 - s0);
    out[(long long)c * row_len + e] = (double)mean;
}

// ------------------------------------------------------------- column sums ------
constexpr int CS_ROWS = 512;
template <typename T>
__global__ __launch_bounds__(256) void colsum64_stage1(const T* __restrict__ X, long long ldx, long long n, int d,
                                                       double* __restrict__ part) {
    __shared__ double sh[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.y * CS_ROWS;
    const long long r1 = r0 + CS_ROWS < n ? r0 + CS_ROWS : n;
    double a = 0.0;
    if (c < d)
        for (long long r = r0 + q; r < r1; r += 4) a += (double)X[r * ldx + c];
    sh[q][threadIdx.x & 63] = a;
    __syncthreads();
    if (q == 0 && c < d) {
        const int l = threadIdx.x;
        part[(long long)blockIdx.y * d + c] = (sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l]);
    }
}
__global__ void colsum64_stage2(const double* __restrict__ part, int nparts, int d, double* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d) return;
    double a = 0.0;
    for (int i = 0; i < nparts; ++i) a += part[(long long)i * d + c];
    out[c] = a;
}

// ------------------------------------------------------------- f64 MFMA GEMM ----
constexpr int DM = 64, DN = 64, DK = 16, DLD = 66;

struct Mat64 {
    const void* p;
    long long ld;
    const double* mean;   // indexed by the CONTIGUOUS (storage column) index, or null
    int is_f32;
};

__device__ inline double ld_elem(const Mat64& m, long long off) {
    return m.is_f32 ? (double)reinterpret_cast<const float*>(m.p)[off] : reinterpret_cast<const double*>(m.p)[off];
}

// Stage a (DK x 64) k-major tile  S[k][x]  of a matrix stored either [x][k] (KC) or [k][x].
template <bool KC>
__device__ inline void stage_tile(const Mat64& m, double (*S)[DLD], int x0, int X, int k0, int kend, int tid) {
    if (KC) {
        const int x = x0 + (tid >> 2), kb = k0 + (tid & 3) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = kb + i;
            double v = 0.0;
            if (x < X && k < kend) {
                v = ld_elem(m, (long long)x * m.ld + k);
                if (m.mean) v -= m.mean[k];
            }
            S[(tid & 3) * 4 + i][tid >> 2] = v;
        }
    } else {
        const int k = k0 + (tid >> 4), xb = x0 + (tid & 15) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = xb + i;
            double v = 0.0;
            if (x < X && k < kend) {
                v = ld_elem(m, (long long)k * m.ld + x);
                if (m.mean) v -= m.mean[x];
            }
            S[tid >> 4][(tid & 15) * 4 + i] = v;
        }
    }
}

template <bool AK, bool BK>
__global__ __launch_bounds__(256) void gemm_f64_kernel(Mat64 A, Mat64 B, void* __restrict__ Cp, long long ldc, int c_is_f32,
                                                       int M, int N, int K, int kchunk, long long slab_stride) {
    __shared__ __attribute__((aligned(16))) double As[DK][DLD];
    __shared__ __attribute__((aligned(16))) double Bs[DK][DLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * DM, n0 = blockIdx.x * DN;
    const int kbeg = blockIdx.z * kchunk;
    const int kend = min(K, kbeg + kchunk);

    f64x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f64x4){0.0, 0.0, 0.0, 0.0};

    for (int k0 = kbeg; k0 < kend; k0 += DK) {
        stage_tile<AK>(A, As, m0, M, k0, kend, tid);
        stage_tile<BK>(B, Bs, n0, N, k0, kend, tid);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < DK; ks += 4) {
            const double a = As[ks + kq][wave * 16 + n];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double b = Bs[ks + kq][j * 16 + n];
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D layout (f64 16x16x4): col = lane & 15, row = (lane >> 4) + 4 * i
    const long long zoff = (long long)blockIdx.z * slab_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + wave * 16 + kq + 4 * i, col = n0 + j * 16 + n;
            if (row < M && col < N) {
                const long long o = zoff + (long long)row * ldc + col;
                if (c_is_f32) reinterpret_cast<float*>(Cp)[o] = (float)acc[j][i];
                else reinterpret_cast<double*>(Cp)[o] = acc[j][i];
            }
        }
}

__global__ void slab_reduce64(const double* __restrict__ slabs, int splits, long long slab_stride,
                              double* __restrict__ C, long long ldc, int M, int N) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)M * N) return;
    double s = 0.0;
    for (int z = 0; z < splits; ++z) s += slabs[(long long)z * slab_stride + idx];
    C[(idx / N) * ldc + (idx % N)] = s;
}

int xcov_splits(long long n, int da, int db) {
    const long long tiles = (long long)cdiv(da, DM) * cdiv(db, DN);
    long long s = (1024 + tiles - 1) / tiles;
    const long long maxs = (n + 255) / 256;
    if (s > maxs) s = maxs;
    if (s > 512) s = 512;
    if (s < 1) s = 1;
    return (int)s;
}

template <bool AK, bool BK>
int launch_gemm64(const Mat64& A, const Mat64& B, void* C, long long ldc, int c_is_f32, int M, int N, int K,
                  int splits, int kchunk, long long slab_stride, hipStream_t st) {
    dim3 grid(cdiv(N, DN), cdiv(M, DM), splits);
    hipLaunchKernelGGL((gemm_f64_kernel<AK, BK>), grid, dim3(256), 0, st, A, B, C, ldc, c_is_f32, M, N, K, kchunk,
                       slab_stride);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------ Jacobi ------
// One round of the round-robin ordering: block b rotates the column pair (p, q).
__device__ inline void rr_pair(int ne, int r, int i, int& p, int& q) {
    const int mod = ne - 1;
    if (i == 0) { p = ne - 1; q = r % mod; }
    else { p = (r + i) % mod; q = ((r - i) % mod + mod) % mod; }
    if (p > q) { int t = p; p = q; q = t; }
}

__global__ __launch_bounds__(256) void jacobi_round_kernel(double* __restrict__ W, long long ldw, double* __restrict__ V,
                                                           long long ldv, int m, int n, int ne, int round,
                                                           unsigned long long* __restrict__ off_bits) {
    __shared__ double sa[256], sb[256], sg[256];
    __shared__ double cs[2];
    int p, q;
    rr_pair(ne, round, blockIdx.x, p, q);
    if (q >= n) return;                      // padded (odd n) partner
    double* wp = W + (long long)p * ldw;
    double* wq = W + (long long)q * ldw;
    double a = 0.0, b = 0.0, g = 0.0;
    for (int i = threadIdx.x; i < m; i += 256) {
        const double x = wp[i], y = wq[i];
        a += x * x; b += y * y; g += x * y;
    }
    sa[threadIdx.x] = a; sb[threadIdx.x] = b; sg[threadIdx.x] = g;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sa[threadIdx.x] += sa[threadIdx.x + s];
            sb[threadIdx.x] += sb[threadIdx.x + s];
            sg[threadIdx.x] += sg[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double alpha = sa[0], beta = sb[0], gamma = sg[0];
        double c = 1.0, s = 0.0;
        const double denom = sqrt(alpha * beta);
        const double rel = denom > 0.0 ? fabs(gamma) / denom : 0.0;
        if (rel > 1e-15 && fabs(gamma) > 0.0) {
            const double zeta = (beta - alpha) / (2.0 * gamma);
            const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            c = 1.0 / sqrt(1.0 + t * t);
            s = c * t;
        }
        cs[0] = c; cs[1] = s;
        if (off_bits) atomicMax(off_bits, (unsigned long long)__double_as_longlong(rel));
    }
    __syncthreads();
    const double c = cs[0], s = cs[1];
    if (s == 0.0) return;
    for (int i = threadIdx.x; i < m; i += 256) {
        const double x = wp[i], y = wq[i];
        wp[i] = c * x - s * y;
        wq[i] = s * x + c * y;
    }
    double* vp = V + (long long)p * ldv;
    double* vq = V + (long long)q * ldv;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double x = vp[i], y = vq[i];
        vp[i] = c * x - s * y;
        vq[i] = s * x + c * y;
    }
}

inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

}  // namespace

extern "C" int xps_cnd_avg_f32(const float* data, const int32_t* order, const int32_t* start, double* out,
                               int n_cond, int64_t row_len, void* stream) {
    XPS_CHECK_ARG(data && order && start && out && n_cond >= 0 && row_len >= 0, "bad argument");
    if (n_cond == 0 || row_len == 0) return XPS_OK;
    hipLaunchKernelGGL(cnd_avg_kernel<float>, dim3(cdiv(row_len, 256), n_cond), dim3(256), 0, (hipStream_t)stream,
                       data, (const int*)order, (const int*)start, out, (long long)row_len);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_cnd_avg_f64(const double* data, const int32_t* order, const int32_t* start, double* out,
                               int n_cond, int64_t row_len, void* stream) {
    XPS_CHECK_ARG(data && order && start && out && n_cond >= 0 && row_len >= 0, "bad argument");
    if (n_cond == 0 || row_len == 0) return XPS_OK;
    hipLaunchKernelGGL(cnd_avg_kernel<double>, dim3(cdiv(row_len, 256), n_cond), dim3(256), 0, (hipStream_t)stream,
                       data, (const int*)order, (const int*)start, out, (long long)row_len);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" size_t xps_colsum_f64_workspace(int64_t n, int d) {
    if (n <= 0 || d <= 0) return 16;
    return (size_t)cdiv(n, CS_ROWS) * d * sizeof(double) + 16;
}

extern "C" int xps_colsum_f64(const void* X, int is_f32, int64_t ldx, int64_t n, int d, double* out,
                              void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(X && out && n >= 0 && d >= 1, "bad argument");
    if (workspace_bytes < xps_colsum_f64_workspace(n, d) || !workspace || !aligned8(workspace)) {
        xps_set_error("xps_colsum_f64: workspace too small or misaligned");
        return XPS_E_WORKSPACE;
    }
    const int nparts = n > 0 ? cdiv(n, CS_ROWS) : 0;
    double* part = (double*)workspace;
    if (nparts > 0) {
        if (is_f32)
            hipLaunchKernelGGL(colsum64_stage1<float>, dim3(cdiv(d, 64), nparts), dim3(256), 0, (hipStream_t)stream,
                               (const float*)X, (long long)ldx, (long long)n, d, part);
        else
            hipLaunchKernelGGL(colsum64_stage1<double>, dim3(cdiv(d, 64), nparts), dim3(256), 0, (hipStream_t)stream,
                               (const double*)X, (long long)ldx, (long long)n, d, part);
        XPS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(colsum64_stage2, dim3(cdiv(d, 256)), dim3(256), 0, (hipStream_t)stream, part, nparts, d, out);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" size_t xps_xcov_f64_workspace(int64_t n, int da, int db) {
    if (n <= 0 || da <= 0 || db <= 0) return 16;
    return (size_t)xcov_splits(n, da, db) * da * db * sizeof(double) + 16;
}

extern "C" int xps_xcov_f64(const void* A, int a_is_f32, int64_t lda, const double* mean_a,
                            const void* B, int b_is_f32, int64_t ldb, const double* mean_b,
                            double* C, int64_t ldc, int64_t n, int da, int db,
                            void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(A && B && C && n >= 1 && da >= 1 && db >= 1, "bad argument");
    XPS_CHECK_ARG(n < (1LL << 31), "row count must fit in int32");
    if (workspace_bytes < xps_xcov_f64_workspace(n, da, db) || !workspace || !aligned8(workspace)) {
        xps_set_error("xps_xcov_f64: workspace too small or misaligned");
        return XPS_E_WORKSPACE;
    }
    const int splits = xcov_splits(n, da, db);
    const int kchunk = ((cdiv(n, splits) + DK - 1) / DK) * DK;
    Mat64 a{A, (long long)lda, mean_a, a_is_f32}, b{B, (long long)ldb, mean_b, b_is_f32};
    const long long slab = (long long)da * db;
    if (launch_gemm64<false, false>(a, b, workspace, db, 0, da, db, (int)n, splits, kchunk, slab, (hipStream_t)stream)) {
        xps_set_error("xps_xcov_f64: launch failed");
        return XPS_E_HIP;
    }
    hipLaunchKernelGGL(slab_reduce64, dim3(cdiv(slab, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const double*)workspace, splits, slab, C, (long long)ldc, da, db);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_apply_f64(const void* X, int x_is_f32, int64_t ldx, const double* mean, const double* W,
                             int64_t ldw, void* Y, int y_is_f32, int64_t ldy, int64_t n, int d_in, int d_out,
                             void* stream) {
    XPS_CHECK_ARG(X && W && Y && n >= 0 && d_in >= 1 && d_out >= 1, "bad argument");
    XPS_CHECK_ARG(n < (1LL << 31), "row count must fit in int32");
    if (n == 0) return XPS_OK;
    Mat64 a{X, (long long)ldx, mean, x_is_f32}, b{W, (long long)ldw, nullptr, 0};
    const int kchunk = ((d_in + DK - 1) / DK) * DK + DK;
    if (launch_gemm64<true, false>(a, b, Y, ldy, y_is_f32, (int)n, d_out, d_in, 1, kchunk, 0, (hipStream_t)stream)) {
        xps_set_error("xps_apply_f64: launch failed");
        return XPS_E_HIP;
    }
    return XPS_OK;
}

extern "C" int xps_dgemm_small(const double* A, int64_t lda, int ta, const double* B, int64_t ldb, int tb,
                               double* C, int64_t ldc, int M, int N, int K, void* stream) {
    XPS_CHECK_ARG(A && B && C && M >= 0 && N >= 0 && K >= 0, "bad argument");
    if (M == 0 || N == 0) return XPS_OK;
    Mat64 a{A, (long long)lda, nullptr, 0}, b{B, (long long)ldb, nullptr, 0};
    const int kchunk = ((K + DK - 1) / DK) * DK + DK;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // op(A) = A  : stored [m][k] (k contiguous);  op(A) = A^T : stored [k][m]
    // op(B) = B  : stored [k][n];                 op(B) = B^T : stored [n][k] (k contiguous)
    if (!ta && !tb) rc = launch_gemm64<true, false>(a, b, C, ldc, 0, M, N, K, 1, kchunk, 0, st);
    else if (!ta && tb) rc = launch_gemm64<true, true>(a, b, C, ldc, 0, M, N, K, 1, kchunk, 0, st);
    else if (ta && !tb) rc = launch_gemm64<false, false>(a, b, C, ldc, 0, M, N, K, 1, kchunk, 0, st);
    else rc = launch_gemm64<false, true>(a, b, C, ldc, 0, M, N, K, 1, kchunk, 0, st);
    if (rc) {
        xps_set_error("xps_dgemm_small: launch failed");
        return XPS_E_HIP;
    }
    return XPS_OK;
}

extern "C" size_t xps_jacobi_f64_workspace(int n) { (void)n; return 16; }

extern "C" int xps_jacobi_sweeps_f64(double* W, int64_t ldw, double* V, int64_t ldv, int m, int n, int sweeps,
                                     double* off, void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    XPS_CHECK_ARG(W && V && m >= 1 && n >= 1 && sweeps >= 0, "bad argument");
    XPS_CHECK_ARG(ldw >= m && ldv >= n, "leading dimensions too small (column-major)");
    hipStream_t st = (hipStream_t)stream;
    const int ne = (n + 1) & ~1;
    if (n == 1 || sweeps == 0) {
        if (off && hipMemsetAsync(off, 0, sizeof(double), st) != hipSuccess) return XPS_E_HIP;
        return XPS_OK;
    }
    for (int s = 0; s < sweeps; ++s) {
        const bool last = (s == sweeps - 1);
        if (last && off) {
            if (hipMemsetAsync(off, 0, sizeof(double), st) != hipSuccess) {
                xps_set_error("xps_jacobi_sweeps_f64: memset failed");
                return XPS_E_HIP;
            }
        }
        for (int r = 0; r < ne - 1; ++r) {
            hipLaunchKernelGGL(jacobi_round_kernel, dim3(ne / 2), dim3(256), 0, st, W, (long long)ldw, V,
                               (long long)ldv, m, n, ne, r, (last && off) ? (unsigned long long*)off : nullptr);
        }
        XPS_CHECK_LAUNCH();
    }
    return XPS_OK;
}
