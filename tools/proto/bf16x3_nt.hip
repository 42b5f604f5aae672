// Prototype: C (M,N) f32 = A (M,K) f32 · B (N,K)^T f32 computed on the bf16 matrix pipe with a 3-product split
// (a = ah + al, b = bh + bl; ah*bh + ah*bl + al*bh, fp32 accumulate): ~2^-16 relative product error.
// Requires M % 128 == 0, N % 128 == 0, K % 32 == 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int BM = 128, BN = 128;

__device__ inline void split4(const f32x4 x, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 h = (__bf16)x[j];
        hi[j] = h;
        lo[j] = (__bf16)(x[j] - (float)h);
    }
}

template <int NPROD, int BK>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16split(const float* __restrict__ A, const float* __restrict__ B,
                                                             float* __restrict__ C, const float* __restrict__ bias,
                                                             int M, int N, int K) {
    constexpr int ROWB = BK * 2 + 16;            // bytes per LDS row: BK bf16 + 16 B pad
    constexpr int ARR = BM * ROWB, STAGE = 4 * ARR;
    constexpr int KQ = BK / 4;                   // k quads per row
    constexpr int RP = 256 / KQ;                 // rows per pass
    constexpr int NV = BM / RP;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = N / BN;
    const int bm = blockIdx.x / ntn, bn = blockIdx.x % ntn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // staging: vec v = tid + 256 i -> row v / 8, k-quad v % 8
    f32x4 ra[NV], rb[NV];
    const int srow = tid / KQ, skq = tid % KQ;
    const float* Ap = A + (long long)(m0 + srow) * K + 4 * skq;
    const float* Bp = B + (long long)(n0 + srow) * K + 4 * skq;
    const long long rstep = (long long)RP * K;
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            ra[i] = *(const f32x4*)(Ap + i * rstep + k0);
            rb[i] = *(const f32x4*)(Bp + i * rstep + k0);
        }
    };
    auto lstore = [&](int buf) {
        unsigned char* base = lds + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int off = (srow + RP * i) * ROWB + skq * 8;
            bf16x4 hi, lo;
            split4(ra[i], hi, lo);
            *(bf16x4*)(base + off) = hi;
            *(bf16x4*)(base + ARR + off) = lo;
            split4(rb[i], hi, lo);
            *(bf16x4*)(base + 2 * ARR + off) = hi;
            *(bf16x4*)(base + 3 * ARR + off) = lo;
        }
    };
    const int nk = K / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const unsigned char* base = lds + buf * STAGE;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int offa = (wm * 64 + i * 32 + r) * ROWB + s * 32 + h * 16;
                ah[i] = *(const bf16x8*)(base + offa);
                al[i] = *(const bf16x8*)(base + ARR + offa);
                const int offb = (wn * 64 + i * 32 + r) * ROWB + s * 32 + h * 16;
                bh[i] = *(const bf16x8*)(base + 2 * ARR + offb);
                bl[i] = *(const bf16x8*)(base + 3 * ARR + offb);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // small terms first
                    if (NPROD >= 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    // epilogue: C[m][n], A rows -> output rows.  MFMA computes D[row i][col j] = sum_k A[i][k] B[k][j]; the B operand
    // fragment here is B^T's column = our B row n, so D col = n.  lane&31 = col, rows in registers.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + r;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                C[(long long)row * N + col] = acc[i][j][e] + bv;
            }
        }
}
}  // namespace

extern "C" int proto_gemm_nt_bf16split(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                                       int nprod, int bk, void* stream) {
    if (M % BM || N % BN || K % bk) return -1;
    const int blocks = (M / BM) * (N / BN);
#define LAUNCH(NP, BKV) { const size_t shm = 2 * 4 * BM * (BKV * 2 + 16); \
        (void)hipFuncSetAttribute((const void*)gemm_nt_bf16split<NP, BKV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
        hipLaunchKernelGGL((gemm_nt_bf16split<NP, BKV>), dim3(blocks), dim3(256), shm, (hipStream_t)stream, A, B, C, bias, M, N, K); }
    if (nprod == 3 && bk == 32) LAUNCH(3, 32)
    else if (nprod == 3 && bk == 16) LAUNCH(3, 16)
    else if (nprod == 3 && bk == 64) LAUNCH(3, 64)
    else if (nprod == 1) LAUNCH(1, 32)
    else return -3;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
