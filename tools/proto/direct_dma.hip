// Prototype harness for the direct-form LDS-DMA k loop of csrc/xps_gemm_dma.h (direct_dma_pipeline): NT (C = A B^T) and NN (C = A B) on
// split4 operands against the register-staged loop of xps_gemm_big.h (bits + time), stand-alone.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o direct_dma tools/proto/direct_dma.hip && ./direct_dma [nt|nn] [M N K]
#include "../../cross_patient_speech_decoding_amd/csrc/xps_gemm_dma.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

void xps_set_error(const char*, ...) {}
int xps_internal_gemm_mode() { return 1; }
using namespace xps_big;
extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];

__global__ void split4_inplace(float* p, long long n4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n4; i += (long long)gridDim.x * blockDim.x) reinterpret_cast<f32x4*>(p)[i] = split4_pack(reinterpret_cast<f32x4*>(p)[i]);
}

template <bool BKX, bool DMA>
__global__ __launch_bounds__(512, 2) void direct_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb,
                                                         float* __restrict__ C, int N, int K) {
    const int tiles_n = N / TN;
    const int lid = xps_tile::xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * TM, n0 = (lid % tiles_n) * TN;
    f32x16 acc[4][2];
    big_zero(acc);
    if constexpr (DMA) {
        direct_dma_pipeline<BKX>(acc, A, lda, B, ldb, m0, n0, 0, K / 32, smem_dyn);
    } else {
        BigStage& st = *reinterpret_cast<BigStage*>(smem_dyn);
        BigLoader<true> la;
        BigLoader<!BKX> lb;
        la.init(A, lda, m0, 0, threadIdx.x);
        lb.init(B, ldb, n0, 0, threadIdx.x);
        f32x4 cs = {0.f, 0.f, 0.f, 0.f};
        big_pipeline_t<true, !BKX, true, true>(acc, cs, false, la, lb, K / 16, st);
    }
#ifdef ABL_NOSTORE        // (timing ablation: the k loop without the C stores; the accumulators are kept alive)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(acc[i][j]));
    if (threadIdx.x == 9999) big_store_c(acc, C, N, nullptr, m0, n0, 0);
#else
    big_store_c(acc, C, N, nullptr, m0, n0, 0);
#endif
}

static float randn() {
    float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = (rand() + 1.f) / (RAND_MAX + 2.f);
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

template <bool BKX, bool DMA>
static float run(const float* A, long long lda, const float* B, long long ldb, float* C, int M, int N, int K, int reps) {
    const int lds = DMA ? direct_dma_lds<BKX>() : (int)sizeof(BigStage);
    hipFuncSetAttribute(reinterpret_cast<const void*>(direct_kernel<BKX, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = (M / TM) * (N / TN);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((direct_kernel<BKX, DMA>), dim3(blocks), dim3(512), lds, 0, A, lda, B, ldb, C, N, K);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((direct_kernel<BKX, DMA>), dim3(blocks), dim3(512), lds, 0, A, lda, B, ldb, C, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("launch error: %s\n", hipGetErrorString(e)); exit(2); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

template <bool BKX>
static int test(int M, int N, int K) {
    printf("%s %d x %d x %d (%d blocks), LDS %d B\n", BKX ? "NN" : "NT", M, N, K, (M / 256) * (N / 256), direct_dma_lds<BKX>());
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    srand(11);
    for (auto& x : hA) x = randn();
    for (auto& x : hB) x = randn() * 0.5f;
    float *A, *B, *C0, *C1;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4);
    const size_t cn = (size_t)M * N;
    hipMalloc(&C0, cn * 4); hipMalloc(&C1, cn * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split4_inplace, dim3(2048), dim3(256), 0, 0, A, (long long)hA.size() / 4);
    hipLaunchKernelGGL(split4_inplace, dim3(2048), dim3(256), 0, 0, B, (long long)hB.size() / 4);
    hipMemset(C0, 0xff, cn * 4); hipMemset(C1, 0xff, cn * 4);
    const long long ldb = BKX ? N : K;                    // NN: B is [K][N]; NT: B is [N][K]
    const float t0 = run<BKX, false>(A, K, B, ldb, C0, M, N, K, 10);
    const float t1 = run<BKX, true>(A, K, B, ldb, C1, M, N, K, 10);
    hipDeviceSynchronize();
    std::vector<float> c0(cn), c1(cn);
    hipMemcpy(c0.data(), C0, cn * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c1.data(), C1, cn * 4, hipMemcpyDeviceToHost);
    size_t diff = 0; double maxd = 0;
    for (size_t i = 0; i < cn; ++i) if (memcmp(&c0[i], &c1[i], 4)) { ++diff; maxd = fmax(maxd, fabs((double)c0[i] - c1[i])); }
    double worst = 0;
    for (int s = 0; s < 64; ++s) {
        const int i = (s * 977 + 13) % M, j = (s * 613 + 5) % N;
        double ref = 0, sab = 0;
        for (int k = 0; k < K; ++k) {
            const double b = BKX ? hB[(size_t)k * N + j] : hB[(size_t)j * K + k];
            ref += (double)hA[(size_t)i * K + k] * b; sab += fabs((double)hA[(size_t)i * K + k] * b);
        }
        worst = fmax(worst, fabs(c1[(size_t)i * N + j] - ref) / sab);
    }
    const double flop = 2.0 * M * N * K;
    printf("  register-staged split4: %8.1f us  %6.1f TF  issued frac %.3f\n", t0, flop / t0 / 1e6, 3 * flop / t0 / 1e6 / 2500e0);
    printf("  LDS-DMA               : %8.1f us  %6.1f TF  issued frac %.3f\n", t1, flop / t1 / 1e6, 3 * flop / t1 / 1e6 / 2500e0);
    printf("  bitwise: %zu of %zu elements differ (max |d| %.3e); error vs fp64 / sum|ab|: %.3e (bound 1.6e-5)\n", diff, cn, maxd, worst);
#ifdef XPS_DMA_STAMP
    {
        std::vector<unsigned long long> st(4096 * 4);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_dma_stamp), st.size() * 8);
        const int nw = 4096;
        double sum[4] = {0, 0, 0, 0};
        for (int w = 0; w < nw; ++w) for (int j = 0; j < 4; ++j) sum[j] += (double)st[w * 4 + j];
        const double nk = (double)(K / 32) * nw;
        printf("  stamps, cycles per 32-deep stage and wave: vmcnt wait %.0f, two barriers %.0f, DMA issue (8 pieces) %.0f, reads + 48 MFMAs %.0f (floor 1536 per wave, 3072 per SIMD)\n",
               sum[0] / nk, sum[1] / nk, sum[2] / nk, sum[3] / nk);
    }
#endif
    hipFree(A); hipFree(B); hipFree(C0); hipFree(C1);
    return (diff || worst > 1.6e-5) ? 3 : 0;
}

int main(int argc, char** argv) {
    const bool nn = argc > 1 && !strcmp(argv[1], "nn");
    const int M = argc > 2 ? atoi(argv[2]) : 32768, N = argc > 3 ? atoi(argv[3]) : (nn ? 1024 : 1536), K = argc > 4 ? atoi(argv[4]) : (nn ? 1536 : 1024);
    if (M % 256 || N % 256 || K % 32) { printf("bad shape\n"); return 1; }
    return nn ? test<true>(M, N, K) : test<false>(M, N, K);
}
