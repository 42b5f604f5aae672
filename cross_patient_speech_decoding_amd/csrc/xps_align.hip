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
// one 1024-thread block per 64 columns: sixteen partial-row groups with 8 loads in flight each, folded in a fixed order
// (a single thread walking hundreds of partials serially took 76 us)
__global__ __launch_bounds__(1024) void colsum64_stage2(const double* __restrict__ part, int nparts, int d, double* __restrict__ out) {
    __shared__ double sh[16][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + l;
    double a = 0.0;
    if (c < d) {
        int i = q;
        for (; i + 7 * 16 < nparts; i += 8 * 16) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long long)(i + 16 * u) * d + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; i < nparts; i += 16) a += part[(long long)i * d + c];
    }
    sh[q][l] = a;
    __syncthreads();
    if (q == 0 && c < d) {
        a = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += sh[k][l];
        out[c] = a;
    }
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

// Rotation angle of one column pair from (alpha, beta, gamma) = (|p|^2, |q|^2, p.q); rel = |cos angle|.
__device__ inline void jacobi_angle(double alpha, double beta, double gamma, double& c, double& s, double& rel) {
    c = 1.0; s = 0.0;
    const double denom = sqrt(alpha * beta);
    rel = denom > 0.0 ? fabs(gamma) / denom : 0.0;
    if (rel > 1e-15 && fabs(gamma) > 0.0) {
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        c = 1.0 / sqrt(1.0 + t * t);
        s = c * t;
    }
}

__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One launch per round; block b owns pair (p, q).  EPT > 0: both columns (m <= 256*EPT) stay in registers
// between the dot products and the rotation (one pass over W instead of two); EPT == 0: any m, two passes.
// V == nullptr: the caller does not need the accumulated rotations (eigenvectors of a positive-definite
// matrix are the normalised columns of W) -> half the traffic.
template <int EPT>
__global__ __launch_bounds__(256) void jacobi_round_kernel(double* __restrict__ W, long long ldw, double* __restrict__ V,
                                                           long long ldv, int m, int n, int ne, int round,
                                                           unsigned long long* __restrict__ off_bits) {
    __shared__ double red[3][4];
    int p, q;
    rr_pair(ne, round, blockIdx.x, p, q);
    if (q >= n) return;                      // padded (odd n) partner
    double* wp = W + (long long)p * ldw;
    double* wq = W + (long long)q * ldw;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double x[EPT > 0 ? EPT : 1], y[EPT > 0 ? EPT : 1];
    double a = 0.0, b = 0.0, g = 0.0;
    if (EPT > 0) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + 256 * e;
            x[e] = i < m ? wp[i] : 0.0;
            y[e] = i < m ? wq[i] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) { a += x[e] * x[e]; b += y[e] * y[e]; g += x[e] * y[e]; }
    } else {
        for (int i = tid; i < m; i += 256) {
            const double xv = wp[i], yv = wq[i];
            a += xv * xv; b += yv * yv; g += xv * yv;
        }
    }
    a = wave_sum(a); b = wave_sum(b); g = wave_sum(g);
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = g; }
    __syncthreads();
    const double alpha = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double beta = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double gamma = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    double c, s, rel;
    jacobi_angle(alpha, beta, gamma, c, s, rel);
    if (tid == 0 && off_bits) atomicMax(off_bits, (unsigned long long)__double_as_longlong(rel));
    if (s == 0.0) return;
    if (EPT > 0) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int i = tid + 256 * e;
            if (i < m) {
                wp[i] = c * x[e] - s * y[e];
                wq[i] = s * x[e] + c * y[e];
            }
        }
    } else {
        for (int i = tid; i < m; i += 256) {
            const double xv = wp[i], yv = wq[i];
            wp[i] = c * xv - s * yv;
            wq[i] = s * xv + c * yv;
        }
    }
    if (V) {
        double* vp = V + (long long)p * ldv;
        double* vq = V + (long long)q * ldv;
        for (int i = tid; i < n; i += 256) {
            const double xv = vp[i], yv = vq[i];
            vp[i] = c * xv - s * yv;
            vq[i] = s * xv + c * yv;
        }
    }
}

// Block one-sided Jacobi for large n without V.  The per-pair kernel above streams the whole matrix through HBM /
// Infinity Cache once per round (n - 1 rounds per sweep: 8 MB x 1023 at n = 1024, bandwidth-bound at ~8 us a round).
// Here the columns are grouped in blocks of JB = 8; a workgroup loads TWO blocks (16 columns, 128 KB of LDS),
// rotates all JB x JB cross pairs in JB conflict-free inner rounds (wave w keeps column I_w in registers and meets
// J_{(w+r) mod JB} in inner round r), and writes the blocks back: one pass over the matrix per JB inner rounds, and
// n/JB - 1 launches per sweep instead of n - 1.  The pairs INSIDE a block are rotated by one extra launch per
// sweep (WITHIN mode: blocks 2j and 2j+1, round-robin over the 8 columns of each).  Every pair is visited exactly
// once per sweep, so the sweep counts are those of the cyclic orderings.
constexpr int JB = 8, JB_MAXM = 1024, JB_EPL = JB_MAXM / 64;
template <bool CROSS>
__global__ __launch_bounds__(512) void jacobi_block_kernel(double* __restrict__ W, long long ldw, int m, int n, int nb,
                                                           int round, unsigned long long* __restrict__ off_bits) {
    extern __shared__ double jcols[];                      // [2 * JB][m]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int I, J;
    if (CROSS) rr_pair(nb, round, blockIdx.x, I, J);
    else { I = 2 * blockIdx.x; J = I + 1; }
    const int base[2] = {I * JB, J * JB};
    if (base[0] >= n && base[1] >= n) return;
    if (CROSS && (base[0] >= n || base[1] >= n)) return;   // phantom partner: nothing to rotate
    // load the (up to) 16 columns; wave w loads columns w and w + 8
    for (int c = wave; c < 2 * JB; c += 8) {
        const int g = base[c / JB] + (c % JB);
        double* dst = jcols + (size_t)c * m;
        if (g < n) {
            const double* src = W + (long long)g * ldw;
            for (int i = lane; i < m; i += 64) dst[i] = src[i];
        }
    }
    __syncthreads();
    double offmax = 0.0;
    if (CROSS) {
        const int gi = base[0] + wave;                    // this wave's resident column (block I)
        const bool have_x = gi < n;
        double x[JB_EPL];
#pragma unroll
        for (int e = 0; e < JB_EPL; ++e) {
            const int i = lane + 64 * e;
            x[e] = (have_x && i < m) ? jcols[(size_t)wave * m + i] : 0.0;
        }
        for (int r = 0; r < JB; ++r) {
            const int pj = (wave + r) % JB;
            const bool act = have_x && (base[1] + pj < n);
            double* yc = jcols + (size_t)(JB + pj) * m;
            if (act) {
                double y[JB_EPL];
                double a = 0.0, b = 0.0, g = 0.0;
#pragma unroll
                for (int e = 0; e < JB_EPL; ++e) {
                    const int i = lane + 64 * e;
                    y[e] = i < m ? yc[i] : 0.0;
                    a += x[e] * x[e]; b += y[e] * y[e]; g += x[e] * y[e];
                }
                a = wave_sum(a); b = wave_sum(b); g = wave_sum(g);
                double c, s, rel;
                jacobi_angle(a, b, g, c, s, rel);
                offmax = fmax(offmax, rel);
                if (s != 0.0) {
#pragma unroll
                    for (int e = 0; e < JB_EPL; ++e) {
                        const int i = lane + 64 * e;
                        const double xv = x[e], yv = y[e];
                        x[e] = c * xv - s * yv;
                        if (i < m) yc[i] = s * xv + c * yv;
                    }
                }
            }
            __syncthreads();
        }
        if (have_x) {
#pragma unroll
            for (int e = 0; e < JB_EPL; ++e) {
                const int i = lane + 64 * e;
                if (i < m) jcols[(size_t)wave * m + i] = x[e];
            }
        }
    } else {
        const int half = wave >> 2, pw = wave & 3;        // waves 0-3: pairs inside block I, 4-7: inside block J
        for (int r = 0; r < JB - 1; ++r) {
            int p, q;
            rr_pair(JB, r, pw, p, q);
            const bool act = base[half] + q < n;           // p < q
            double* pc = jcols + (size_t)(half * JB + p) * m;
            double* qc = jcols + (size_t)(half * JB + q) * m;
            if (act) {
                double x[JB_EPL], y[JB_EPL];
                double a = 0.0, b = 0.0, g = 0.0;
#pragma unroll
                for (int e = 0; e < JB_EPL; ++e) {
                    const int i = lane + 64 * e;
                    x[e] = i < m ? pc[i] : 0.0;
                    y[e] = i < m ? qc[i] : 0.0;
                    a += x[e] * x[e]; b += y[e] * y[e]; g += x[e] * y[e];
                }
                a = wave_sum(a); b = wave_sum(b); g = wave_sum(g);
                double c, s, rel;
                jacobi_angle(a, b, g, c, s, rel);
                offmax = fmax(offmax, rel);
                if (s != 0.0) {
#pragma unroll
                    for (int e = 0; e < JB_EPL; ++e) {
                        const int i = lane + 64 * e;
                        if (i < m) {
                            pc[i] = c * x[e] - s * y[e];
                            qc[i] = s * x[e] + c * y[e];
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int c = wave; c < 2 * JB; c += 8) {
        const int g = base[c / JB] + (c % JB);
        if (g < n) {
            double* dst = W + (long long)g * ldw;
            const double* src = jcols + (size_t)c * m;
            for (int i = lane; i < m; i += 64) dst[i] = src[i];
        }
    }
    if (off_bits && lane == 0 && offmax > 0.0) atomicMax(off_bits, (unsigned long long)__double_as_longlong(offmax));
}

// Whole decomposition of a SMALL matrix (n <= 128 columns) in ONE launch: one workgroup per matrix of the
// batch, W (and V when it fits) live in LDS, 8 lanes per column pair, all sweeps and the convergence test run
// inside the kernel (the per-round launch version spends ~6 us per round: 6 ms for a 128 x 128 eigenproblem).
// LDS columns are padded to an odd length so that the 8 pairs of a wave fall on different banks.
// V mode: 0 = no V, 1 = V in LDS, 2 = V in global memory (n x n does not fit beside W).
constexpr int JS_THREADS = 512, JS_LPP = 8, JS_MAX_N = 2 * (JS_THREADS / JS_LPP), JS_LDS_DOUBLES = 19200;
constexpr int JS_EPL = JS_MAX_N / JS_LPP;          // elements of a column per lane (m, n <= JS_MAX_N)
__global__ __launch_bounds__(JS_THREADS) void jacobi_small_kernel(double* __restrict__ Wg, long long ldw, long long sw,
                                                                  double* __restrict__ Vg, long long ldv, long long sv,
                                                                  int m, int n, int vmode, int max_sweeps, double tol,
                                                                  int* __restrict__ sweeps_done, double* __restrict__ off_out) {
    extern __shared__ double jlds[];
    __shared__ double s_off[JS_THREADS / 64];
    const int tid = threadIdx.x;
    const int ldW = m | 1, ldVs = n | 1;
    double* W = jlds;
    Wg += (long long)blockIdx.x * sw;
    if (Vg) Vg += (long long)blockIdx.x * sv;
    double* V = vmode == 1 ? jlds + (size_t)n * ldW : Vg;
    const long long ldV = vmode == 1 ? (long long)ldVs : ldv;
    for (int idx = tid; idx < n * m; idx += JS_THREADS) {
        const int col = idx / m, i = idx - col * m;
        W[col * ldW + i] = Wg[(long long)col * ldw + i];
    }
    if (vmode)
        for (int idx = tid; idx < n * n; idx += JS_THREADS) {
            const int col = idx / n, i = idx - col * n;
            V[col * ldV + i] = col == i ? 1.0 : 0.0;
        }
    __syncthreads();
    const int ne = (n + 1) & ~1, pairs = ne / 2;
    const int t = tid & (JS_LPP - 1), pj = tid / JS_LPP;
    int sweep = 0;
    double last_off = 0.0;
    while (sweep < max_sweeps) {
        double offmax = 0.0;
        for (int r = 0; r < ne - 1; ++r) {
            if (pj < pairs) {
                int p, q;
                rr_pair(ne, r, pj, p, q);
                if (q < n) {
                    double* wp = W + p * ldW;
                    double* wq = W + q * ldW;
                    // both columns in registers (<= 16 elements per lane): one LDS read per element and no
                    // read-after-write stalls between the dot products and the rotation
                    double x[JS_EPL], y[JS_EPL];
                    double a = 0.0, b = 0.0, g = 0.0;
#pragma unroll
                    for (int e = 0; e < JS_EPL; ++e) {
                        const int i = t + JS_LPP * e;
                        x[e] = i < m ? wp[i] : 0.0;
                        y[e] = i < m ? wq[i] : 0.0;
                    }
#pragma unroll
                    for (int e = 0; e < JS_EPL; ++e) { a += x[e] * x[e]; b += y[e] * y[e]; g += x[e] * y[e]; }
#pragma unroll
                    for (int o = JS_LPP / 2; o > 0; o >>= 1) {
                        a += __shfl_xor(a, o, JS_LPP); b += __shfl_xor(b, o, JS_LPP); g += __shfl_xor(g, o, JS_LPP);
                    }
                    double c, s, rel;
                    jacobi_angle(a, b, g, c, s, rel);
                    offmax = fmax(offmax, rel);
                    if (s != 0.0) {
#pragma unroll
                        for (int e = 0; e < JS_EPL; ++e) {
                            const int i = t + JS_LPP * e;
                            if (i < m) {
                                wp[i] = c * x[e] - s * y[e];
                                wq[i] = s * x[e] + c * y[e];
                            }
                        }
                        if (vmode) {
                            double* vp = V + p * ldV;
                            double* vq = V + q * ldV;
#pragma unroll
                            for (int e = 0; e < JS_EPL; ++e) {           // all loads in flight before the first store
                                const int i = t + JS_LPP * e;
                                x[e] = i < n ? vp[i] : 0.0;
                                y[e] = i < n ? vq[i] : 0.0;
                            }
#pragma unroll
                            for (int e = 0; e < JS_EPL; ++e) {
                                const int i = t + JS_LPP * e;
                                if (i < n) {
                                    vp[i] = c * x[e] - s * y[e];
                                    vq[i] = s * x[e] + c * y[e];
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        ++sweep;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) offmax = fmax(offmax, __shfl_xor(offmax, o, 64));
        if ((tid & 63) == 0) s_off[tid >> 6] = offmax;
        __syncthreads();
        double mx = 0.0;
#pragma unroll
        for (int w = 0; w < JS_THREADS / 64; ++w) mx = fmax(mx, s_off[w]);
        last_off = mx;
        __syncthreads();
        if (mx <= tol) break;                   // block-uniform
    }
    for (int idx = tid; idx < n * m; idx += JS_THREADS) {
        const int col = idx / m, i = idx - col * m;
        Wg[(long long)col * ldw + i] = W[col * ldW + i];
    }
    if (vmode == 1)
        for (int idx = tid; idx < n * n; idx += JS_THREADS) {
            const int col = idx / n, i = idx - col * n;
            Vg[(long long)col * ldv + i] = V[col * ldV + i];
        }
    if (tid == 0) {
        if (sweeps_done) sweeps_done[blockIdx.x] = sweep;
        if (off_out) off_out[blockIdx.x] = last_off;
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
    hipLaunchKernelGGL(colsum64_stage2, dim3(cdiv(d, 64)), dim3(1024), 0, (hipStream_t)stream, part, nparts, d, out);
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

// ---- skinny float64 products: few output tiles, long contraction (the subspace iteration of the MCCA eigensolve: a 1024 x 1024
// matrix against a 1024 x 45 block was ONE launch of 16 blocks, 80-110 us) -> split-K slabs + one reduce, deterministic ----
static int dgemm_splits(int M, int N, int K) {
    const long long tiles = (long long)cdiv(M, DM) * cdiv(N, DN);
    long long s = (256 + tiles - 1) / tiles;
    const long long maxs = (K + 4 * DK - 1) / (4 * DK);
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return (int)s;
}

// out = alpha * sum_z slab_z + beta * Y + gamma * Vp   (Vp may be null); element-wise, row-major with leading dimension N
static __global__ void slab_reduce64_axpy(const double* __restrict__ slabs, int splits, long long slab_stride, const double* __restrict__ Y,
                                   const double* __restrict__ Vp, double* __restrict__ out, long long total, double alpha, double beta,
                                   double gamma) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    double s = 0.0;
    for (int z = 0; z < splits; ++z) s += slabs[(long long)z * slab_stride + idx];
    double v = alpha * s;
    if (Y) v += beta * Y[idx];
    if (Vp) v += gamma * Vp[idx];
    out[idx] = v;
}

extern "C" size_t xps_dgemm_splitk_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 16;
    return (size_t)dgemm_splits(M, N, K) * M * N * sizeof(double) + 16;
}

// C = op(A) op(B) like xps_dgemm_small, the contraction split over workgroups (deterministic slabs in `workspace`)
extern "C" int xps_dgemm_splitk(const double* A, int64_t lda, int ta, const double* B, int64_t ldb, int tb, double* C, int64_t ldc,
                                int M, int N, int K, void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(A && B && C && M >= 1 && N >= 1 && K >= 1, "bad argument");
    if (!workspace || workspace_bytes < xps_dgemm_splitk_workspace(M, N, K) || !aligned8(workspace)) {
        xps_set_error("xps_dgemm_splitk: workspace too small or misaligned");
        return XPS_E_WORKSPACE;
    }
    const int splits = dgemm_splits(M, N, K);
    const int kchunk = ((cdiv(K, splits) + DK - 1) / DK) * DK;
    Mat64 a{A, (long long)lda, nullptr, 0}, b{B, (long long)ldb, nullptr, 0};
    hipStream_t st = (hipStream_t)stream;
    const long long slab = (long long)M * N;
    int rc;
    if (!ta && !tb) rc = launch_gemm64<true, false>(a, b, workspace, N, 0, M, N, K, splits, kchunk, slab, st);
    else if (!ta && tb) rc = launch_gemm64<true, true>(a, b, workspace, N, 0, M, N, K, splits, kchunk, slab, st);
    else if (ta && !tb) rc = launch_gemm64<false, false>(a, b, workspace, N, 0, M, N, K, splits, kchunk, slab, st);
    else rc = launch_gemm64<false, true>(a, b, workspace, N, 0, M, N, K, splits, kchunk, slab, st);
    if (rc) { xps_set_error("xps_dgemm_splitk: launch failed"); return XPS_E_HIP; }
    hipLaunchKernelGGL(slab_reduce64, dim3(cdiv(slab, 256)), dim3(256), 0, st, (const double*)workspace, splits, slab, C, (long long)ldc, M, N);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

// Chebyshev filter of a block (Zhou & Saad's scaled three-term recurrence; alignment/_linalg.py: eigh_sym_top):
//   Y_1 = (C A - c A) sigma_1 / e ;  Y_{j+1} = (C Y_j - c Y_j) 2 sigma_{j+1} / e - sigma_j sigma_{j+1} Y_{j-1},
//   sigma_{j+1} = 1 / (2 / sigma_1 - sigma_j), deg products in all, ONE host call: per product a split-K launch and a reduce that
// applies the recurrence (the Python loop paid ~0.2 ms of host + launch time per product, 129 products per 8-view MCCA fit).
// C: n x n symmetric (leading dimension ldc); A, out: n x m row-major (leading dimension m); out may not alias A.
extern "C" size_t xps_cheb_filter_f64_workspace(int n, int m) {
    if (n <= 0 || m <= 0) return 16;
    return ((size_t)dgemm_splits(n, m, n) + 3) * n * m * sizeof(double) + 16;
}

extern "C" int xps_cheb_filter_f64(const double* C, int64_t ldc, int n, const double* A, int m, int deg, double c, double e, double sigma1,
                                   double* out, void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(C && A && out && n >= 1 && m >= 1 && deg >= 1 && e > 0.0 && out != A, "bad argument");
    if (!workspace || workspace_bytes < xps_cheb_filter_f64_workspace(n, m) || !aligned8(workspace)) {
        xps_set_error("xps_cheb_filter_f64: workspace too small or misaligned");
        return XPS_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int splits = dgemm_splits(n, m, n);
    const int kchunk = ((cdiv(n, splits) + DK - 1) / DK) * DK;
    const long long slab = (long long)n * m;
    double* slabs = (double*)workspace;
    double* ring[3] = {slabs + (long long)splits * slab, slabs + (long long)(splits + 1) * slab, slabs + (long long)(splits + 2) * slab};
    Mat64 cm{C, (long long)ldc, nullptr, 0};
    const double* Y = A;
    const double* Vp = nullptr;
    double sigma = sigma1;
    for (int j = 0; j < deg; ++j) {
        double alpha, gamma = 0.0;
        if (j == 0) alpha = sigma1 / e;
        else {
            const double sigma2 = 1.0 / (2.0 / sigma1 - sigma);
            alpha = 2.0 * sigma2 / e;
            gamma = -sigma * sigma2;
            sigma = sigma2;
        }
        Mat64 ym{Y, (long long)m, nullptr, 0};
        if (launch_gemm64<true, false>(cm, ym, slabs, m, 0, n, m, n, splits, kchunk, slab, st)) {
            xps_set_error("xps_cheb_filter_f64: launch failed");
            return XPS_E_HIP;
        }
        double* dst = (j == deg - 1) ? out : ring[j % 3];
        hipLaunchKernelGGL(slab_reduce64_axpy, dim3(cdiv(slab, 256)), dim3(256), 0, st, (const double*)slabs, splits, slab, Y, Vp, dst, slab,
                           alpha, -c * alpha, gamma);
        Vp = Y;
        Y = dst;
    }
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

// One Lanczos step behind the split-K product slabs = C v (no reorthogonalisation; alignment/_linalg.py: _lanczos_bounds):
//   w = sum_z slab_z - beta_prev vp;  a = w . v;  w -= a v;  b = ||w||;  vp = v;  v = w / b;  alpha[j] = a, beta[j] = b.
// ONE workgroup (n is at most a few thousand): fixed reduction order, deterministic.
static __global__ __launch_bounds__(1024) void lanczos_step_kernel(const double* __restrict__ slabs, int splits, int n, double* __restrict__ v,
                                                                   double* __restrict__ vp, double* __restrict__ w, double* __restrict__ alpha,
                                                                   double* __restrict__ beta, int j) {
    __shared__ double red[16];
    __shared__ double bc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto block_sum = [&](double x) -> double {
        x = wave_sum(x);
        __syncthreads();
        if (lane == 0) red[wave] = x;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < 16; ++i) s += red[i];
            bc = s;
        }
        __syncthreads();
        return bc;
    };
    const double bprev = j > 0 ? beta[j - 1] : 0.0;
    double dot = 0.0;
    for (int i = tid; i < n; i += 1024) {
        double s = 0.0;
        for (int z = 0; z < splits; ++z) s += slabs[(long long)z * n + i];
        s -= bprev * vp[i];
        w[i] = s;
        dot += s * v[i];
    }
    const double a = block_sum(dot);
    double nn = 0.0;
    for (int i = tid; i < n; i += 1024) {
        const double x = w[i] - a * v[i];
        w[i] = x;
        nn += x * x;
    }
    const double b = sqrt(block_sum(nn));
    for (int i = tid; i < n; i += 1024) {
        vp[i] = v[i];
        v[i] = w[i] / b;
    }
    if (tid == 0) { alpha[j] = a; beta[j] = b; }
}

extern "C" size_t xps_lanczos_f64_workspace(int n) {
    if (n <= 0) return 16;
    return ((size_t)dgemm_splits(n, 1, n) + 3) * n * sizeof(double) + 16;
}

// `steps` Lanczos steps on the symmetric n x n matrix C from the start vector v0 (normalised here): alpha[steps], beta[steps]
// (device arrays) = the diagonal / off-diagonal of the tridiagonal matrix; enqueued by this one call (2 launches per step).
extern "C" int xps_lanczos_f64(const double* C, int64_t ldc, int n, int steps, const double* v0, double* alpha, double* beta,
                               void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(C && v0 && alpha && beta && n >= 1 && steps >= 1, "bad argument");
    if (!workspace || workspace_bytes < xps_lanczos_f64_workspace(n) || !aligned8(workspace)) {
        xps_set_error("xps_lanczos_f64: workspace too small or misaligned");
        return XPS_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int splits = dgemm_splits(n, 1, n);
    const int kchunk = ((cdiv(n, splits) + DK - 1) / DK) * DK;
    double* slabs = (double*)workspace;
    double* v = slabs + (long long)splits * n;
    double* vp = v + n;
    double* w = vp + n;
    if (hipMemcpyAsync(v, v0, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemsetAsync(vp, 0, (size_t)n * sizeof(double), st) != hipSuccess) {
        xps_set_error("xps_lanczos_f64: copy failed");
        return XPS_E_HIP;
    }
    Mat64 cm{C, (long long)ldc, nullptr, 0};
    for (int j = 0; j < steps; ++j) {
        Mat64 vm{v, 1, nullptr, 0};
        if (launch_gemm64<true, false>(cm, vm, slabs, 1, 0, n, 1, n, splits, kchunk, (long long)n, st)) {
            xps_set_error("xps_lanczos_f64: launch failed");
            return XPS_E_HIP;
        }
        hipLaunchKernelGGL(lanczos_step_kernel, dim3(1), dim3(1024), 0, st, (const double*)slabs, splits, n, v, vp, w, alpha, beta, j);
    }
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" size_t xps_jacobi_f64_workspace(int n) { (void)n; return 16; }

namespace {
void launch_jacobi_round(double* W, long long ldw, double* V, long long ldv, int m, int n, int ne, int r,
                         unsigned long long* off, hipStream_t st) {
    if (m <= 1024)
        hipLaunchKernelGGL(jacobi_round_kernel<4>, dim3(ne / 2), dim3(256), 0, st, W, ldw, V, ldv, m, n, ne, r, off);
    else if (m <= 2048)
        hipLaunchKernelGGL(jacobi_round_kernel<8>, dim3(ne / 2), dim3(256), 0, st, W, ldw, V, ldv, m, n, ne, r, off);
    else
        hipLaunchKernelGGL(jacobi_round_kernel<0>, dim3(ne / 2), dim3(256), 0, st, W, ldw, V, ldv, m, n, ne, r, off);
}

// 0 = not supported, else the V mode the small kernel would use
int jacobi_small_vmode(int m, int n, int want_v) {
    if (m < 1 || n < 1 || n > JS_MAX_N || m > JS_MAX_N) return -1;
    const long long w = (long long)n * (m | 1);
    if (w > JS_LDS_DOUBLES) return -1;
    if (!want_v) return 0;
    return (w + (long long)n * (n | 1) <= JS_LDS_DOUBLES) ? 1 : 2;
}
// ---------------------------------------------------------------------------------------------------------------------
// Whitening factor of a batch of small symmetric positive definite blocks (the regularised within-view covariances R_b of
// the MCCA eigenproblem, AlignMCCA._gevp): A = scale * R + shift * I = L L^T (right-looking Cholesky), S = L^-T (upper
// triangular), so that S^T A S = I.  One workgroup per block, A in LDS (rows padded to an odd length); the inverse is found
// column by column (thread c solves L x = e_c; the loops over (i, k) are uniform, so L[i][k] is one broadcast read and
// x_c[k] -- kept in the unused upper triangle -- a conflict-free one).  info[b] = 0, or j + 1 when pivot j is not positive.
constexpr int CW_THREADS = 256, CW_MAX_N = 136;
__global__ __launch_bounds__(CW_THREADS) void chol_whiten_kernel(const double* __restrict__ Rg, long long ldr, long long sr,
                                                                 double scale, double shift, double* __restrict__ Ag,
                                                                 long long lda, long long sa, double* __restrict__ Sg,
                                                                 long long ldsg, long long ss, int n, int* __restrict__ info) {
    extern __shared__ double cw[];
    __shared__ int s_bad;
    const int tid = threadIdx.x, ld = n | 1;
    double* A = cw;                       // n x ld
    double* dinv = cw + (size_t)n * ld;   // 1 / L[j][j]
    Rg += (long long)blockIdx.x * sr;
    Sg += (long long)blockIdx.x * ss;
    if (Ag) Ag += (long long)blockIdx.x * sa;
    if (tid == 0) s_bad = 0;
    for (int idx = tid; idx < n * n; idx += CW_THREADS) {
        const int i = idx / n, j = idx - i * n;
        // the block is symmetric up to rounding of its two GEMM halves: the LOWER triangle is the one factorised
        double v = scale * Rg[(long long)i * ldr + j] + (i == j ? shift : 0.0);
        A[i * ld + j] = v;
        if (Ag) Ag[(long long)i * lda + j] = v;
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        const double d = A[j * ld + j];
        if (!(d > 0.0) || !(d < 1.7e308)) {              // uniform: every thread reads the same pivot
            if (tid == 0) { s_bad = j + 1; }
            break;
        }
        const double r = 1.0 / sqrt(d);
        __syncthreads();                                  // pivot read by all before column j is scaled
        for (int i = j + 1 + tid; i < n; i += CW_THREADS) A[i * ld + j] *= r;
        if (tid == 0) { A[j * ld + j] = sqrt(d); dinv[j] = r; }
        __syncthreads();
        // trailing update of the lower triangle: A[i][k] -= L[i][j] * L[k][j], j < k <= i
        const int rem = n - 1 - j;
        const int cnt = rem * (rem + 1) / 2;
        for (int t = tid; t < cnt; t += CW_THREADS) {
            // t -> (a, b), 0 <= b <= a < rem (row-wise enumeration of the triangle)
            int a = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while ((a + 1) * (a + 2) / 2 <= t) ++a;
            while (a * (a + 1) / 2 > t) --a;
            const int b = t - a * (a + 1) / 2;
            const int i = j + 1 + a, k = j + 1 + b;
            A[i * ld + k] -= A[i * ld + j] * A[k * ld + j];
        }
        __syncthreads();
    }
    __syncthreads();
    if (s_bad) {
        if (tid == 0 && info) info[blockIdx.x] = s_bad;
        return;
    }
    if (tid == 0 && info) info[blockIdx.x] = 0;
    // inverse: thread c owns column c of X = L^-1, stored at A[c][i] (i > c: the upper triangle), X[c][c] = dinv[c]
    if (tid < n) {
        const int c = tid;
        for (int i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int k = 0; k < i; ++k) {
                const double l = A[i * ld + k];                                  // broadcast
                const double x = (k == c) ? dinv[c] : A[c * ld + k];             // x_c[k]; zero for k < c (masked below)
                acc += (k >= c) ? l * x : 0.0;
            }
            if (i > c) A[c * ld + i] = -acc * dinv[i];
        }
    }
    __syncthreads();
    // S = L^-T = X^T: S[r][c] = X[c][r] = A[r][c] for c > r, dinv[r] on the diagonal, zero below
    for (int idx = tid; idx < n * n; idx += CW_THREADS) {
        const int r = idx / n, c = idx - r * n;
        Sg[(long long)r * ldsg + c] = c > r ? A[r * ld + c] : (c == r ? dinv[r] : 0.0);
    }
}
}  // namespace

extern "C" int xps_chol_whiten_supported(int n) { return n >= 1 && n <= CW_MAX_N; }

extern "C" int xps_chol_whiten_f64(const double* R, int64_t ldr, int64_t stride_r, double scale, double shift, double* A,
                                   int64_t lda, int64_t stride_a, double* S, int64_t lds_, int64_t stride_s, int n, int batch,
                                   int32_t* info, void* stream) {
    XPS_CHECK_ARG(R && S && batch >= 0 && ldr >= n && lds_ >= n && (!A || lda >= n), "bad argument");
    if (batch == 0) return XPS_OK;
    if (!xps_chol_whiten_supported(n)) {
        xps_set_error("xps_chol_whiten_f64: n = %d does not fit one workgroup's LDS (at most %d)", n, CW_MAX_N);
        return XPS_E_INVALID;
    }
    const size_t lds = ((size_t)n * (n | 1) + n) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(chol_whiten_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(((size_t)CW_MAX_N * (CW_MAX_N | 1) + CW_MAX_N) * sizeof(double))) != hipSuccess) {
            xps_set_error("xps_chol_whiten_f64: cannot raise the dynamic LDS limit");
            return XPS_E_HIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(chol_whiten_kernel, dim3(batch), dim3(CW_THREADS), lds, (hipStream_t)stream, R, (long long)ldr,
                       (long long)stride_r, scale, shift, A, (long long)lda, (long long)stride_a, S, (long long)lds_,
                       (long long)stride_s, n, (int*)info);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_jacobi_small_supported(int m, int n, int want_v) { return jacobi_small_vmode(m, n, want_v) >= 0; }

extern "C" int xps_jacobi_small_f64(double* W, int64_t ldw, int64_t stride_w, double* V, int64_t ldv, int64_t stride_v,
                                    int m, int n, int batch, int max_sweeps, double tol, int32_t* sweeps_done,
                                    double* off, void* stream) {
    XPS_CHECK_ARG(W && m >= 1 && n >= 1 && batch >= 0 && max_sweeps >= 0, "bad argument");
    XPS_CHECK_ARG(ldw >= m && (!V || ldv >= n), "leading dimensions too small (column-major)");
    if (batch == 0) return XPS_OK;
    const int vmode = jacobi_small_vmode(m, n, V != nullptr);
    if (vmode < 0) {
        xps_set_error("xps_jacobi_small_f64: %d x %d does not fit the single-workgroup kernel", m, n);
        return XPS_E_INVALID;
    }
    const size_t lds = ((size_t)n * (m | 1) + (vmode == 1 ? (size_t)n * (n | 1) : 0)) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                JS_LDS_DOUBLES * (int)sizeof(double)) != hipSuccess) {
            xps_set_error("xps_jacobi_small_f64: cannot raise the dynamic LDS limit");
            return XPS_E_HIP;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(jacobi_small_kernel, dim3(batch), dim3(JS_THREADS), lds, (hipStream_t)stream, W, (long long)ldw,
                       (long long)stride_w, V, (long long)ldv, (long long)stride_v, m, n, vmode, max_sweeps, tol,
                       (int*)sweeps_done, off);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_jacobi_sweeps_f64(double* W, int64_t ldw, double* V, int64_t ldv, int m, int n, int sweeps,
                                     double* off, void* workspace, size_t workspace_bytes, void* stream) {
    (void)workspace; (void)workspace_bytes;
    XPS_CHECK_ARG(W && m >= 1 && n >= 1 && sweeps >= 0, "bad argument");
    XPS_CHECK_ARG(ldw >= m && (!V || ldv >= n), "leading dimensions too small (column-major)");
    hipStream_t st = (hipStream_t)stream;
    const int ne = (n + 1) & ~1;
    if (n == 1 || sweeps == 0) {
        if (off && hipMemsetAsync(off, 0, sizeof(double), st) != hipSuccess) return XPS_E_HIP;
        return XPS_OK;
    }
    const bool block_path = (V == nullptr) && m <= JB_MAXM && n > 4 * JB;
    const int nb = ((n + JB - 1) / JB + 1) & ~1;         // blocks, padded to an even count
    const size_t block_lds = (size_t)2 * JB * m * sizeof(double);
    if (block_path) {
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_block_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * JB * JB_MAXM * (int)sizeof(double)) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(jacobi_block_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * JB * JB_MAXM * (int)sizeof(double)) != hipSuccess) {
                xps_set_error("xps_jacobi_sweeps_f64: cannot raise the dynamic LDS limit");
                return XPS_E_HIP;
            }
            attr_set = true;
        }
    }
    for (int s = 0; s < sweeps; ++s) {
        const bool last = (s == sweeps - 1);
        if (last && off) {
            if (hipMemsetAsync(off, 0, sizeof(double), st) != hipSuccess) {
                xps_set_error("xps_jacobi_sweeps_f64: memset failed");
                return XPS_E_HIP;
            }
        }
        if (block_path) {
            unsigned long long* ob = (last && off) ? (unsigned long long*)off : nullptr;
            hipLaunchKernelGGL(jacobi_block_kernel<false>, dim3(nb / 2), dim3(512), block_lds, st, W, (long long)ldw, m, n, nb,
                               0, ob);
            for (int r = 0; r < nb - 1; ++r)
                hipLaunchKernelGGL(jacobi_block_kernel<true>, dim3(nb / 2), dim3(512), block_lds, st, W, (long long)ldw, m, n,
                                   nb, r, ob);
            XPS_CHECK_LAUNCH();
            continue;
        }
        for (int r = 0; r < ne - 1; ++r) {
            launch_jacobi_round(W, (long long)ldw, V, (long long)ldv, m, n, ne, r,
                                (last && off) ? (unsigned long long*)off : nullptr, st);
        }
        XPS_CHECK_LAUNCH();
    }
    return XPS_OK;
}
