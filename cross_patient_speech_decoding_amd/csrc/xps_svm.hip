// Batched C-SVC training on a precomputed kernel matrix for gfx950: the bagged linear one-vs-one SVMs of the config-1 decode
// path (reference: decoders/cross_pt_decoders.py:11-86 hand the pooled features to sklearn's SVC(kernel='linear') inside a
// BaggingClassifier, scripts/aligned_decode_svm.py:262-263; sklearn wraps libsvm 3.x).  One workgroup per binary problem
// (class pair) runs libsvm's SMO iteration -- first-order choice of i, second-order choice of j (WSS 2 of Fan, Chen & Lin
// 2005), the clipped two-variable update, the gradient update, the eps stopping rule and calculate_rho -- on alpha / gradient
// vectors kept in LDS; the rows of Q it needs are gathered from the Gram matrix K = X X^T (computed once per fit by the f64
// MFMA GEMM).  No shrinking and no kernel cache (K is resident): the iterates follow libsvm's rules, the sums run in another
// order, so alpha agrees to rounding, not bit for bit.  Latency-bound small-problem code: a problem of n points costs
// O(n) iterations of five barriers.
#include "xps_common.h"

namespace {

constexpr int SVM_THREADS = 256;
constexpr double SVM_TAU = 1e-12;

struct ArgD { double v; int i; };

// block-wide arg-max over (value, index); ties -> the LARGER index (libsvm updates on '>=' while scanning upwards)
__device__ inline ArgD block_argmax(ArgD a, ArgD* scratch) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v = __shfl_xor(a.v, o);
        const int i = __shfl_xor(a.i, o);
        if (v > a.v || (v == a.v && i > a.i)) { a.v = v; a.i = i; }
    }
    if (lane == 0) scratch[wave] = a;
    __syncthreads();
    ArgD r = scratch[0];
#pragma unroll
    for (int w = 1; w < SVM_THREADS / 64; ++w) {
        const ArgD b = scratch[w];
        if (b.v > r.v || (b.v == r.v && b.i > r.i)) r = b;
    }
    __syncthreads();
    return r;
}

__device__ inline double block_max(double a, double* scratch) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmax(a, __shfl_xor(a, o));
    if (lane == 0) scratch[wave] = a;
    __syncthreads();
    double r = scratch[0];
#pragma unroll
    for (int w = 1; w < SVM_THREADS / 64; ++w) r = fmax(r, scratch[w]);
    __syncthreads();
    return r;
}

// problem p: points idx[off[p] .. off[p + 1]), the first npos[p] of them carry y = +1, the others y = -1
__global__ __launch_bounds__(SVM_THREADS) void svm_smo_kernel(const double* __restrict__ K, long long ldk, const int* __restrict__ idx,
                                                              const int* __restrict__ off, const int* __restrict__ npos, const double* __restrict__ cbound,
                                                              double eps, int max_iter, double* __restrict__ alpha_out,
                                                              double* __restrict__ rho_out, int* __restrict__ iters_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int p = blockIdx.x, tid = threadIdx.x;
    const int o0 = off[p], n = off[p + 1] - o0, np = npos[p];
    double* alpha = reinterpret_cast<double*>(smem_raw);
    double* G = alpha + n;
    double* Cb = G + n;                                    // per-point upper bound C * sample_weight (libsvm: C[i] = W[i] * C)
    int* id = reinterpret_cast<int*>(Cb + n);
    __shared__ ArgD s_arg[SVM_THREADS / 64];
    __shared__ double s_red[SVM_THREADS / 64];
    __shared__ double s_da[2];
    for (int t = tid; t < n; t += SVM_THREADS) { alpha[t] = 0.0; G[t] = -1.0; Cb[t] = cbound[o0 + t]; id[t] = idx[o0 + t]; }
    __syncthreads();
    auto yv = [&](int t) { return t < np ? 1.0 : -1.0; };
    const double NEG = -1e300;
    int it = 0;
    for (; it < max_iter; ++it) {
        // i: maximal violating element of I_up
        ArgD best = {NEG, -1};
        for (int t = tid; t < n; t += SVM_THREADS) {
            double v = NEG;
            if (t < np) { if (alpha[t] < Cb[t]) v = -G[t]; }
            else { if (alpha[t] > 0.0) v = G[t]; }
            if (v > best.v || (v == best.v && t > best.i)) { best.v = v; best.i = t; }
        }
        const ArgD gi = block_argmax(best, s_arg);
        const int i = gi.i;
        const double Gmax = gi.v;
        if (i < 0) break;
        const double yi = yv(i);
        const double* Ki = K + (long long)id[i] * ldk;
        const double QDi = Ki[id[i]];
        // j: second-order choice in I_low; Gmax2 for the stopping rule
        ArgD bj = {NEG, -1};                 // maximise -obj_diff = grad_diff^2 / quad
        double g2 = NEG;
        for (int t = tid; t < n; t += SVM_THREADS) {
            const double a = alpha[t], g = G[t];
            const double Kit = Ki[id[t]];
            const double Qit = yi * yv(t) * Kit;
            double grad_diff = -1.0, quad = 1.0;
            bool cand = false;
            if (t < np) {
                if (a > 0.0) { grad_diff = Gmax + g; g2 = fmax(g2, g); quad = QDi + K[(long long)id[t] * ldk + id[t]] - 2.0 * yi * Qit; cand = true; }
            } else {
                if (a < Cb[t]) { grad_diff = Gmax - g; g2 = fmax(g2, -g); quad = QDi + K[(long long)id[t] * ldk + id[t]] + 2.0 * yi * Qit; cand = true; }
            }
            if (cand && grad_diff > 0.0) {
                const double q = quad > 0.0 ? quad : SVM_TAU;
                const double gain = grad_diff * grad_diff / q;
                if (gain > bj.v || (gain == bj.v && t > bj.i)) { bj.v = gain; bj.i = t; }
            }
        }
        const double Gmax2 = block_max(g2, s_red);
        const ArgD gj = block_argmax(bj, s_arg);
        const int j = gj.i;
        if (Gmax + Gmax2 < eps || j < 0) break;
        const double yj = yv(j);
        const double* Kj = K + (long long)id[j] * ldk;
        if (tid == 0) {
            const double QDj = Kj[id[j]];
            const double Qij = yi * yj * Ki[id[j]];
            double ai = alpha[i], aj = alpha[j];
            const double oi = ai, oj = aj;
            const double Ci = Cb[i], Cj = Cb[j];
            if (yi != yj) {
                double quad = QDi + QDj + 2.0 * Qij;
                if (quad <= 0.0) quad = SVM_TAU;
                const double delta = (-G[i] - G[j]) / quad;
                const double diff = ai - aj;
                ai += delta; aj += delta;
                if (diff > 0.0) { if (aj < 0.0) { aj = 0.0; ai = diff; } }
                else { if (ai < 0.0) { ai = 0.0; aj = -diff; } }
                if (diff > Ci - Cj) { if (ai > Ci) { ai = Ci; aj = Ci - diff; } }
                else { if (aj > Cj) { aj = Cj; ai = Cj + diff; } }
            } else {
                double quad = QDi + QDj - 2.0 * Qij;
                if (quad <= 0.0) quad = SVM_TAU;
                const double delta = (G[i] - G[j]) / quad;
                const double sum = ai + aj;
                ai -= delta; aj += delta;
                if (sum > Ci) { if (ai > Ci) { ai = Ci; aj = sum - Ci; } }
                else { if (aj < 0.0) { aj = 0.0; ai = sum; } }
                if (sum > Cj) { if (aj > Cj) { aj = Cj; ai = sum - Cj; } }
                else { if (ai < 0.0) { ai = 0.0; aj = sum; } }
            }
            alpha[i] = ai; alpha[j] = aj;
            s_da[0] = ai - oi; s_da[1] = aj - oj;
        }
        __syncthreads();
        const double dai = s_da[0], daj = s_da[1];
        for (int t = tid; t < n; t += SVM_THREADS) {
            const double yt = yv(t);
            G[t] += yi * yt * Ki[id[t]] * dai + yj * yt * Kj[id[t]] * daj;
        }
        __syncthreads();
    }
    // rho (libsvm calculate_rho) by thread 0: n is small, the kernel is latency-bound anyway
    if (tid == 0) {
        double ub = 1e300, lb = -1e300, sum_free = 0.0;
        int nfree = 0;
        for (int t = 0; t < n; ++t) {
            const double yG = yv(t) * G[t];
            if (alpha[t] >= Cb[t]) { if (t >= np) ub = fmin(ub, yG); else lb = fmax(lb, yG); }
            else if (alpha[t] <= 0.0) { if (t < np) ub = fmin(ub, yG); else lb = fmax(lb, yG); }
            else { ++nfree; sum_free += yG; }
        }
        rho_out[p] = nfree > 0 ? sum_free / nfree : (ub + lb) / 2.0;
        iters_out[p] = it;
    }
    for (int t = tid; t < n; t += SVM_THREADS) alpha_out[o0 + t] = alpha[t];
}

}  // namespace

namespace {
// RBF kernel matrix from a Gram matrix: K[i][j] = exp(-gamma (na[i] + nb[j] - 2 G[i][j])) -- libsvm's own formula
// (svm.cpp Kernel::kernel_rbf: x_square[i] + x_square[j] - 2 dot(x[i], x[j])), element-wise on the device
__global__ __launch_bounds__(256) void rbf_from_gram_kernel(const double* __restrict__ G, long long ldg, const double* __restrict__ na,
                                                            const double* __restrict__ nb, int m, int n, double gamma,
                                                            double* __restrict__ K, long long ldk) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)m * n) return;
    const int i = (int)(idx / n), j = (int)(idx % n);
    K[(long long)i * ldk + j] = exp(-gamma * (na[i] + nb[j] - 2.0 * G[(long long)i * ldg + j]));
}

}  // namespace

extern "C" int xps_rbf_from_gram_f64(const double* G, int64_t ldg, const double* na, const double* nb, int m, int n, double gamma,
                                     double* K, int64_t ldk, void* stream) {
    XPS_CHECK_ARG(G && na && nb && K && m >= 0 && n >= 0 && ldg >= n && ldk >= n && gamma >= 0.0, "bad argument");
    if (m == 0 || n == 0) return XPS_OK;
    hipLaunchKernelGGL(rbf_from_gram_kernel, dim3(cdiv((long long)m * n, 256)), dim3(256), 0, (hipStream_t)stream, G, (long long)ldg, na, nb, m, n,
                       gamma, K, (long long)ldk);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

// the launch asks for max_points * 28 + 64 bytes of dynamic LDS beside ~112 bytes of static LDS (budgeted as 128) under the
// limit of 160 KiB - 1 KiB raised below: the largest problem that LAUNCHES (5813 / 5814 passed the check and failed at launch)
constexpr size_t SVM_LDS_LIMIT = 160 * 1024 - 1024, SVM_LDS_FIXED = 64 + 128;
extern "C" size_t xps_svm_smo_f64_max_points(void) { return (SVM_LDS_LIMIT - SVM_LDS_FIXED) / 28; }

extern "C" int xps_svm_smo_f64(const double* K, int64_t ldk, const int* idx, const int* off, const int* npos, int nprob, int max_points,
                               const double* cbound, double eps, int max_iter, double* alpha, double* rho, int* iters, void* stream) {
    XPS_CHECK_ARG(K && idx && off && npos && cbound && alpha && rho && iters, "null argument");
    XPS_CHECK_ARG(nprob >= 0 && eps > 0.0 && max_iter > 0, "bad parameter");
    XPS_CHECK_ARG(max_points >= 1 && (size_t)max_points <= xps_svm_smo_f64_max_points(), "a binary problem exceeds the LDS-resident limit");
    if (nprob == 0) return XPS_OK;
    const int lds = max_points * 28 + 64;
    static const bool ok = hipFuncSetAttribute((const void*)svm_smo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SVM_LDS_LIMIT) == hipSuccess;
    if (!ok && lds > 64 * 1024) { xps_set_error("xps_svm_smo_f64: cannot raise the dynamic LDS limit"); return XPS_E_HIP; }
    hipLaunchKernelGGL(svm_smo_kernel, dim3(nprob), dim3(SVM_THREADS), lds, (hipStream_t)stream, K, (long long)ldk, idx, off, npos, cbound, eps,
                       max_iter, alpha, rho, iters);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}
