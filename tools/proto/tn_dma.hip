// Prototype harness for csrc/xps_gemm_dma.h: the LDS-DMA k loop of the 256-tile weight-gradient GEMM against the
// register-staged loop of xps_gemm_big.h on the same split4 operands (bits + time), stand-alone (no torch, no libxps).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DXPS_DMA_SHIFT8=0] -o tn_dma tools/proto/tn_dma.hip && ./tn_dma [M N K splits]
#include "../../cross_patient_speech_decoding_amd/csrc/xps_gemm_dma.h"
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

void xps_set_error(const char*, ...) {}
int xps_internal_gemm_mode() { return 1; }

using namespace xps_big;
#ifndef STAMP_MODE
#define STAMP_MODE 1
#endif
extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];

__global__ void split4_inplace(float* p, long long n4) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 v = reinterpret_cast<f32x4*>(p)[i];
        reinterpret_cast<f32x4*>(p)[i] = split4_pack(v);
    }
}

template <int MODE>   // 0: register-staged baseline, 1: DMA, 2: DMA + column sums, 3: baseline + column sums
__global__ __launch_bounds__(512, 2) void tn_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ B, long long ldb,
                                                     float* __restrict__ C, float* __restrict__ CS, int M, int N, int kchunk) {
    const int tiles_n = N / TN, ntiles = (M / TM) * tiles_n;
    const int lid = xps_tile::xcd_remap(blockIdx.x, gridDim.x);
    const int tile = lid % ntiles, z = lid / ntiles;
    const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TN;
    const int kbeg = z * kchunk;
    f32x16 acc[4][2];
    big_zero(acc);
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    f32x16 cacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) cacc[r] = 0.f;
    const int per = (8 + tiles_n - 1) / tiles_n, tn = tile % tiles_n;
    const int cs_lo = min(8, tn * per), cs_hi = min(8, cs_lo + per);
    if constexpr (MODE == 1 || MODE == 2) {
        tn_dma_pipeline<MODE == 2>(acc, cacc, A, lda, B, ldb, m0, n0, kbeg, kchunk / 16, smem_dyn, cs_lo, cs_hi);
    } else {
        BigStage& st = *reinterpret_cast<BigStage*>(smem_dyn);
        BigLoader<false> la, lb;
        la.init(A, lda, m0, kbeg, threadIdx.x);
        lb.init(B, ldb, n0, kbeg, threadIdx.x);
        big_pipeline_t<false, false, true, true>(acc, csum, MODE == 3, la, lb, kchunk / 16, st);
    }
    big_store_c(acc, C + (long long)z * M * N, N, nullptr, m0, n0, 0);
    if (MODE == 2) dma_colsum_store(cacc, CS + (long long)z * M + m0, cs_lo, cs_hi);
    if (MODE == 3 && n0 == 0) {
        float* red = reinterpret_cast<float*>(smem_dyn);
        const int tid = threadIdx.x;
        *reinterpret_cast<f32x4*>(&red[(tid >> 6) * 256 + (tid & 63) * 4]) = csum;
        __syncthreads();
        if (tid < 256) {
            float v = 0.f;
            for (int w = 0; w < 8; ++w) v += red[w * 256 + tid];
            CS[(long long)z * M + m0 + tid] = v;
        }
    }
}

static float randn() {
    float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = (rand() + 1.f) / (RAND_MAX + 2.f);
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

template <int MODE>
static float run(const float* A, const float* B, float* C, float* CS, int M, int N, int K, int splits, int reps) {
    const int lds = (MODE == 1 || MODE == 2) ? DMA_LDS : (int)sizeof(BigStage);
    hipFuncSetAttribute(reinterpret_cast<const void*>(tn_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int blocks = (M / TM) * (N / TN) * splits;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(tn_kernel<MODE>, dim3(blocks), dim3(512), lds, 0, A, (long long)M, B, (long long)N, C, CS, M, N, K / splits);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(tn_kernel<MODE>, dim3(blocks), dim3(512), lds, 0, A, (long long)M, B, (long long)N, C, CS, M, N, K / splits);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("launch error: %s\n", hipGetErrorString(e)); exit(2); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1536, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 40960;
    const int splits = argc > 4 ? atoi(argv[4]) : 10;
    if (M % 256 || N % 256 || K % (16 * splits)) { printf("bad shape\n"); return 1; }
    printf("TN %d x %d x %d, %d k-splits (%d blocks), DMA row shift8 = %d, LDS %d B\n", M, N, K, splits, (M / 256) * (N / 256) * splits, XPS_DMA_SHIFT8, DMA_LDS);
    std::vector<float> hA((size_t)K * M), hB((size_t)K * N);
    srand(7);
    for (auto& x : hA) x = randn();
    for (auto& x : hB) x = randn() * 0.5f;
    float *A, *B, *C0, *C1, *S0, *S1;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4);
    const size_t cn = (size_t)splits * M * N;
    hipMalloc(&C0, cn * 4); hipMalloc(&C1, cn * 4); hipMalloc(&S0, (size_t)splits * M * 4); hipMalloc(&S1, (size_t)splits * M * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split4_inplace, dim3(2048), dim3(256), 0, 0, A, (long long)hA.size() / 4);
    hipLaunchKernelGGL(split4_inplace, dim3(2048), dim3(256), 0, 0, B, (long long)hB.size() / 4);
    hipMemset(C0, 0xff, cn * 4); hipMemset(C1, 0xff, cn * 4);
    const float t0 = run<0>(A, B, C0, S0, M, N, K, splits, 20);
    const float t1 = run<1>(A, B, C1, S1, M, N, K, splits, 20);
    hipDeviceSynchronize();
    std::vector<float> c0(cn), c1(cn);
    hipMemcpy(c0.data(), C0, cn * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c1.data(), C1, cn * 4, hipMemcpyDeviceToHost);
    size_t diff = 0; double maxd = 0;
    for (size_t i = 0; i < cn; ++i) if (memcmp(&c0[i], &c1[i], 4)) { ++diff; maxd = fmax(maxd, fabs((double)c0[i] - c1[i])); }
    // fp64 check of the summed result on a sample of entries (fp32 operands as generated)
    double worst = 0, scale = 0;
    for (int s = 0; s < 64; ++s) {
        const int i = (s * 977 + 13) % M, j = (s * 613 + 5) % N;
        double ref = 0, sab = 0;
        for (int k = 0; k < K; ++k) { ref += (double)hA[(size_t)k * M + i] * hB[(size_t)k * N + j]; sab += fabs((double)hA[(size_t)k * M + i] * hB[(size_t)k * N + j]); }
        double got = 0;
        for (int z = 0; z < splits; ++z) got += c1[(size_t)z * M * N + (size_t)i * N + j];
        worst = fmax(worst, fabs(got - ref) / sab); scale = fmax(scale, sab);
    }
    const double flop = 2.0 * M * N * K;
    printf("baseline (register-staged split4): %8.1f us  %6.1f TF  issued frac %.3f\n", t0, flop / t0 / 1e6, 3 * flop / t0 / 1e6 / 2500e0);
    printf("LDS-DMA                          : %8.1f us  %6.1f TF  issued frac %.3f\n", t1, flop / t1 / 1e6, 3 * flop / t1 / 1e6 / 2500e0);
    printf("bitwise: %zu of %zu elements differ (max |d| %.3e); error vs fp64 / sum|ab|: %.3e (bound 1.6e-5)\n", diff, cn, maxd, worst);
    // column sums
    const float t3 = run<3>(A, B, C0, S0, M, N, K, splits, 5);
    const float t2 = run<2>(A, B, C1, S1, M, N, K, splits, 5);
    hipDeviceSynchronize();
    std::vector<float> s0((size_t)splits * M), s1((size_t)splits * M);
    hipMemcpy(s0.data(), S0, s0.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(s1.data(), S1, s1.size() * 4, hipMemcpyDeviceToHost);
    // column sums: both against fp64 of the fp32 operand values (bound: 2^-16 sum |a|, the split4 rule of tests/test_gpu_split4.py)
    double csw = 0; size_t sd = 0;
    for (int i = 0; i < M; ++i) {
        double ref = 0, sa = 0, g0 = 0, g1 = 0;
        for (int k = 0; k < K; ++k) { ref += hA[(size_t)k * M + i]; sa += fabs(hA[(size_t)k * M + i]); }
        for (int z = 0; z < splits; ++z) { g0 += s0[(size_t)z * M + i]; g1 += s1[(size_t)z * M + i]; }
        csw = fmax(csw, fabs(g1 - ref) / sa);
        if (fabs(g1 - ref) > sa / 65536 + 1e-6 || fabs(g0 - ref) > sa / 65536 + 1e-6) ++sd;
    }
    hipMemcpy(c0.data(), C0, cn * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c1.data(), C1, cn * 4, hipMemcpyDeviceToHost);
    size_t diff2 = 0;
    for (size_t i = 0; i < cn; ++i) diff2 += memcmp(&c0[i], &c1[i], 4) != 0;
    printf("with column sums: baseline %.1f us, DMA %.1f us; column sums out of bound: %zu of %d (DMA worst error / sum|a| %.2e); products differing: %zu\n", t3, t2, sd, M, csw, diff2);
#ifdef XPS_DMA_STAMP
    {
        run<STAMP_MODE>(A, B, C1, S1, M, N, K, splits, 1);
        hipDeviceSynchronize();
        std::vector<unsigned long long> st(4096 * 4);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_dma_stamp), st.size() * 8);
        const int nw = (M / 256) * (N / 256) * splits * 8;
        double sum[4] = {0, 0, 0, 0};
        for (int w = 0; w < nw && w < 4096; ++w) for (int j = 0; j < 4; ++j) sum[j] += (double)st[w * 4 + j];
        const double nk = (double)(K / splits / 16) * (nw < 4096 ? nw : 4096);
        printf("stamps (mode %d), cycles per k-tile and wave: vmcnt wait %.0f, barrier %.0f, DMA issue %.0f, reads + MFMA %.0f (floor 768 per wave, 1536 per SIMD)\n",
               STAMP_MODE, sum[0] / nk, sum[1] / nk, sum[2] / nk, sum[3] / nk);
    }
#endif
    return (diff || sd || diff2 || worst > 1.6e-5) ? 3 : 0;
}
