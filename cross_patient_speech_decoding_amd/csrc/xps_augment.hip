// On-device training-set augmentations (reference nn_models/data_utils/augmentations.py:13-90) for (trial x time x channel)
// fp32 tensors.  All five are single-pass, HBM-bound kernels, coalesced along the channel index (16-byte vectors when the
// channel count allows); the random draw of a call (shift, mask start, scale factor, warp factor, noise tensor) is made by the
// caller exactly as the reference makes it and handed in, so results are reproducible against it:
//   time shift  = torch.roll along time                      (bit-exact)
//   time mask   = zero [start, start + size) along time      (bit-exact)
//   scale       = x * s in fp32                              (bit-exact)
//   jitter      = x + noise * level, two roundings, no fma   (bit-exact)
//   time warp   = scipy.ndimage.zoom(order=1) T -> T2 in double, then torchvision Resize = bilinear antialias resample
//                 T2 -> T (align_corners = False) in fp32, fused: the intermediate is never stored.
#include "xps_common.h"

namespace {

template <bool V4>
__global__ __launch_bounds__(256) void aug_shift_mask_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int T, int C,
                                                             int shift, int mstart, int msize) {
    // out[n, t, :] = (mstart <= t < mstart + msize) ? 0 : x[n, (t - shift) mod T, :]
    const int cw = V4 ? C / 4 : C;
    const long long total = (long long)N * T * cw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cw);
        const long long r = i / cw;
        const int t = (int)(r % T);
        const long long n = r / T;
        int ts = t - shift;
        ts %= T;
        if (ts < 0) ts += T;
        const bool masked = t >= mstart && t < mstart + msize;
        if (V4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!masked) v = reinterpret_cast<const float4*>(x)[(n * T + ts) * cw + c];
            reinterpret_cast<float4*>(out)[i] = v;
        } else {
            out[i] = masked ? 0.f : x[(n * T + ts) * cw + c];
        }
    }
}

__global__ __launch_bounds__(256) void aug_scale_kernel(const float* __restrict__ x, float* __restrict__ out, long long n, float s) {
#pragma clang fp contract(off)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = __fmul_rn(x[i], s);
}

__global__ __launch_bounds__(256) void aug_jitter_kernel(const float* __restrict__ x, const float* __restrict__ noise, float* __restrict__ out,
                                                         long long n, float level) {
    // two roundings like the reference's `data + randn * level` (hipcc contracts even __fadd_rn(a, __fmul_rn(b, c)) into one fma)
#pragma clang fp contract(off)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float scaled = noise[i] * level;
        out[i] = x[i] + scaled;
    }
}

// stage-1 sample u of row (n, :, c): scipy.ndimage.zoom(order = 1, grid_mode = False): coordinate u * (T - 1) / (T2 - 1), double
__device__ inline float zoom_sample(const float* __restrict__ col, long long stride, int T, int T2, int u) {
    if (T2 <= 1 || T <= 1) return col[0];
    const double pos = (double)u * (double)(T - 1) / (double)(T2 - 1);
    int i0 = (int)floor(pos);
    if (i0 > T - 2) i0 = T - 2;
    if (i0 < 0) i0 = 0;
    const double f = pos - (double)i0;
    const double a = (double)col[(long long)i0 * stride], b = (double)col[(long long)(i0 + 1) * stride];
    return (float)((1.0 - f) * a + f * b);
}

__global__ __launch_bounds__(256) void aug_warp_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int T, int C, int T2) {
    // stage 2: torch upsample_bilinear2d_aa along time (T2 -> T), align_corners = False: triangle filter of support max(scale, 1)
    const float scale = (float)T2 / (float)T;
    const float support = scale >= 1.f ? scale : 1.f;
    const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
    const long long total = (long long)N * T * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long r = i / C;
        const int t = (int)(r % T);
        const long long n = r / T;
        const float* col = x + n * T * C + c;
        const float center = scale * ((float)t + 0.5f);
        int xmin = (int)(center - support + 0.5f);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5f);
        if (xmax > T2) xmax = T2;
        float wsum = 0.f;
        for (int j = xmin; j < xmax; ++j) {
            float a = ((float)j - center + 0.5f) * invscale;
            a = a < 0.f ? -a : a;
            wsum += a < 1.f ? 1.f - a : 0.f;
        }
        float acc = 0.f;
        for (int j = xmin; j < xmax; ++j) {
            float a = ((float)j - center + 0.5f) * invscale;
            a = a < 0.f ? -a : a;
            const float w = (a < 1.f ? 1.f - a : 0.f) / wsum;
            acc += w * zoom_sample(col, C, T, T2, j);
        }
        out[i] = acc;
    }
}

inline int blocks_for(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int xps_aug_time_shift_f32(const float* x, float* out, int N, int T, int C, int shift, void* stream) {
    XPS_CHECK_ARG(N >= 0 && T >= 1 && C >= 1, "bad argument");
    if (N == 0) return XPS_OK;
    XPS_CHECK_ARG(x && out, "null argument");
    XPS_CHECK_ARG(x != out, "in-place roll is not supported");
    const bool v4 = C % 4 == 0 && al16(x) && al16(out);
    const long long total = (long long)N * T * (v4 ? C / 4 : C);
    if (v4) hipLaunchKernelGGL(aug_shift_mask_kernel<true>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, out, N, T, C, shift, 0, 0);
    else hipLaunchKernelGGL(aug_shift_mask_kernel<false>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, out, N, T, C, shift, 0, 0);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_aug_time_mask_f32(const float* x, float* out, int N, int T, int C, int start, int size, void* stream) {
    XPS_CHECK_ARG(N >= 0 && T >= 1 && C >= 1, "bad argument");
    XPS_CHECK_ARG(start >= 0 && size >= 0 && start + size <= T, "mask window outside the sequence");
    if (N == 0) return XPS_OK;
    XPS_CHECK_ARG(x && out, "null argument");
    const bool v4 = C % 4 == 0 && al16(x) && al16(out);
    const long long total = (long long)N * T * (v4 ? C / 4 : C);
    if (v4) hipLaunchKernelGGL(aug_shift_mask_kernel<true>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, out, N, T, C, 0, start, size);
    else hipLaunchKernelGGL(aug_shift_mask_kernel<false>, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, x, out, N, T, C, 0, start, size);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_aug_scale_f32(const float* x, float* out, int64_t n, float scale, void* stream) {
    XPS_CHECK_ARG(n >= 0, "bad argument");
    if (n == 0) return XPS_OK;
    XPS_CHECK_ARG(x && out, "null argument");
    hipLaunchKernelGGL(aug_scale_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, out, (long long)n, scale);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_aug_jitter_f32(const float* x, const float* noise, float* out, int64_t n, float level, void* stream) {
    XPS_CHECK_ARG(n >= 0, "bad argument");
    if (n == 0) return XPS_OK;
    XPS_CHECK_ARG(x && noise && out, "null argument");
    hipLaunchKernelGGL(aug_jitter_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, noise, out, (long long)n, level);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_aug_time_warp_f32(const float* x, float* out, int N, int T, int C, int T2, void* stream) {
    XPS_CHECK_ARG(N >= 0 && T >= 1 && C >= 1 && T2 >= 1, "bad argument");
    if (N == 0) return XPS_OK;
    XPS_CHECK_ARG(x && out, "null argument");
    XPS_CHECK_ARG(x != out, "in-place warp is not supported");
    hipLaunchKernelGGL(aug_warp_kernel, dim3(blocks_for((long long)N * T * C)), dim3(256), 0, (hipStream_t)stream, x, out, N, T, C, T2);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}
