// Batch-1 (few-stream) streaming inference kernels for the realtime decoder (BASELINE config 5).
// A 20 ms step touches every weight once and each weight is used by 1..8 streams only, so the step is
// bound by how fast the weights stream through (2.9 MB for the config-5 layer 0), not by FLOPs: one
// WAVE per output row, 16-byte loads along k, shuffle reduction — no LDS, no MFMA.  The GRU cell kernel
// needs no inter-workgroup exchange: hidden unit j depends only on its own rows of W_ih / W_hh and on
// the (shared, read-only) previous state, so one launch per layer suffices and the launches of a step
// are replayed from a hipGraph.
#include "xps_common.h"

namespace {

constexpr int MAXS = 8;   // streams per launch

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int SB>
__device__ inline void dot_rows(const float* __restrict__ wrow, const float* __restrict__ x, int K, int ldx, int lane,
                                float (&acc)[SB]) {
    // wrow: one weight row (K floats, 16-byte aligned when K % 4 == 0); x: SB input rows
    if ((K & 3) == 0) {
        for (int k = lane * 4; k < K; k += 256) {
            const float4 w = *reinterpret_cast<const float4*>(wrow + k);
#pragma unroll
            for (int s = 0; s < SB; ++s) {
                const float4 v = *reinterpret_cast<const float4*>(x + (long long)s * ldx + k);
                acc[s] += w.x * v.x + w.y * v.y + w.z * v.z + w.w * v.w;
            }
        }
    } else {
        for (int k = lane; k < K; k += 64) {
            const float w = wrow[k];
#pragma unroll
            for (int s = 0; s < SB; ++s) acc[s] += w * x[(long long)s * ldx + k];
        }
    }
}

template <int SB>
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                   int N, int K, int B) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    float acc[SB];
#pragma unroll
    for (int s = 0; s < SB; ++s) acc[s] = 0.f;
    dot_rows<SB>(W + (long long)row * K, x, K, K, lane, acc);
#pragma unroll
    for (int s = 0; s < SB; ++s) {
        const float v = wave_sum(acc[s]);
        if (lane == 0 && s < B) out[(long long)s * N + row] = v + (bias ? bias[row] : 0.f);
    }
}

template <int SB>
__global__ __launch_bounds__(256) void gru_cell_gemv_kernel(const float* __restrict__ x, int K,
                                                            const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                            const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                            const float* __restrict__ h_prev, float* __restrict__ h_new,
                                                            int H, int B) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= H) return;
    float gi[3][SB], gh[3][SB];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int s = 0; s < SB; ++s) { gi[g][s] = 0.f; gh[g][s] = 0.f; }
        dot_rows<SB>(w_ih + (long long)(g * H + j) * K, x, K, K, lane, gi[g]);
        dot_rows<SB>(w_hh + (long long)(g * H + j) * H, h_prev, H, H, lane, gh[g]);
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) {
        float v[6];
#pragma unroll
        for (int g = 0; g < 3; ++g) { v[g] = wave_sum(gi[g][s]); v[3 + g] = wave_sum(gh[g][s]); }
        if (lane == 0 && s < B) {
            const float r = sigmoidf_acc(v[0] + b_ih[j] + v[3] + b_hh[j]);
            const float z = sigmoidf_acc(v[1] + b_ih[H + j] + v[4] + b_hh[H + j]);
            const float n = tanhf(v[2] + b_ih[2 * H + j] + r * (v[5] + b_hh[2 * H + j]));
            const float hp = h_prev[(long long)s * H + j];
            h_new[(long long)s * H + j] = n + z * (hp - n);
        }
    }
}

}  // namespace

#define DISPATCH_SB(B, CALL)          \
    if ((B) <= 1) { CALL(1) }         \
    else if ((B) <= 2) { CALL(2) }    \
    else if ((B) <= 4) { CALL(4) }    \
    else { CALL(8) }

extern "C" int xps_gemv_f32(const float* x, const float* W, const float* bias, float* out, int N, int K, int B,
                            void* stream) {
    XPS_CHECK_ARG(x && W && out && N >= 1 && K >= 1, "bad argument");
    XPS_CHECK_ARG(B >= 1 && B <= MAXS, "1..8 streams per call");
    XPS_CHECK_ARG((K & 3) != 0 || (((uintptr_t)x | (uintptr_t)W) & 15) == 0, "x and W must be 16-byte aligned");
    // rows beyond B of x must be readable: callers pass buffers padded to the template width
#define CALL(S) hipLaunchKernelGGL(gemv_kernel<S>, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, x, W, bias, out, N, K, B);
    DISPATCH_SB(B, CALL)
#undef CALL
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_gru_cell_gemv_f32(const float* x, int K, const float* w_ih, const float* w_hh, const float* b_ih,
                                     const float* b_hh, const float* h_prev, float* h_new, int H, int B, void* stream) {
    XPS_CHECK_ARG(x && w_ih && w_hh && b_ih && b_hh && h_prev && h_new && K >= 1 && H >= 1, "bad argument");
    XPS_CHECK_ARG(B >= 1 && B <= MAXS, "1..8 streams per call");
    XPS_CHECK_ARG(h_prev != h_new, "h_new must not alias h_prev (other workgroups still read it)");
    XPS_CHECK_ARG((K & 3) != 0 || (((uintptr_t)x | (uintptr_t)w_ih) & 15) == 0, "x and w_ih must be 16-byte aligned");
    XPS_CHECK_ARG((H & 3) != 0 || (((uintptr_t)h_prev | (uintptr_t)w_hh) & 15) == 0, "h and w_hh must be 16-byte aligned");
#define CALL(S) hipLaunchKernelGGL(gru_cell_gemv_kernel<S>, dim3(cdiv(H, 4)), dim3(256), 0, (hipStream_t)stream, x, K, w_ih, w_hh, b_ih, b_hh, h_prev, h_new, H, B);
    DISPATCH_SB(B, CALL)
#undef CALL
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}
