// Effective MFMA clock of an MI355X under load: every CU runs W waves per SIMD of back-to-back v_mfma_f32_32x32x16_bf16 on
// operands of a chosen bit pattern (zeros / random bf16 / bf16x3-style hi+lo pairs).  N MFMAs per wave, 8 passes (32 cycles)
// each: clock = N * W * 32 / time.  Shows how much of the 2.4 GHz the data-dependent power management leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ __launch_bounds__(512) void mfma_loop(const uint4* __restrict__ ops, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x;
    union { uint4 u; bf16x8 v; } a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i].u = ops[(tid * 6 + i) & 4095];
    for (int i = 0; i < 2; ++i) b[i].u = ops[(tid * 6 + 4 + i) & 4095];
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * blockDim.x + tid] = s;
}

static unsigned short bf16_of(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint4* ops; float* out;
    hipMalloc(&ops, 4096 * 16); hipMalloc(&out, (size_t)cus * 2 * 512 * 4);
    std::vector<unsigned short> h(4096 * 8);
    const char* names[] = {"zeros", "ones", "randn bf16", "small randn (lo-word like, 2^-9 scale)"};
    for (int pat = 0; pat < 4; ++pat) {
        srand(1);
        for (auto& x : h) {
            float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = (rand() + 1.f) / (RAND_MAX + 2.f);
            float g = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
            x = pat == 0 ? 0 : pat == 1 ? bf16_of(1.f) : pat == 2 ? bf16_of(g) : bf16_of(g / 512.f);
        }
        hipMemcpy(ops, h.data(), 4096 * 16, hipMemcpyHostToDevice);
        for (int threads : {256, 512}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(threads), 0, 0, ops, out, iters / 10);
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(threads), 0, 0, ops, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)iters * 24, wps = threads / 256.0;
            const double clk = n * wps * 32 / (ms * 1e-3);
            const double tf = n * (threads / 64.0) * cus * 32768.0 / (ms * 1e-3) / 1e12;
            printf("%-42s %d waves/SIMD: %8.2f ms  %7.1f TFLOP/s (%.3f of 2500)  effective MFMA clock %.2f GHz\n", names[pat], threads / 256, ms, tf, tf / 2500, clk / 1e9);
        }
    }
    return 0;
}
