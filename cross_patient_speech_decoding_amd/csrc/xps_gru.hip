// Fused GRU recurrence for gfx950: h_{t-1} W_hh^T on the f32-input MFMA
// (v_mfma_f32_16x16x4_f32) + sigmoid/tanh gates + hidden update in one persistent
// kernel per layer.  One workgroup owns 16 trials (batch rows) of one direction for
// ALL time steps: its hidden-state tile lives in LDS (double buffered, one barrier
// per step), W_hh streams from L2 straight into MFMA B operands as 16-byte vectors,
// and the input projections gi = x_t W_ih^T + b_ih (one big GEMM, xps_gemm.hip)
// are read once, coalesced along the hidden index.
//
// MFMA operand mapping (16x16x4, lane l: n = l & 15, kq = l >> 4):
//   A[row = n][slot kq], B[slot kq][col = n], D[row = 4*kq + i][col = n] in reg i.
// The four k-steps of a 16-wide k chunk use the permuted assignment
//   slot kq of step s  <->  k = k0 + 4*kq + s
// so one ds_read_b128 of the LDS hidden tile and one 16-byte global load of a W_hh
// row feed four MFMAs.  Gate math is lane-local: a wave computes the r, z and n
// pre-activations of the same 16 hidden units in three accumulators.
//
// PyTorch GRU semantics (gate order r, z, n):
//   r = s(gi_r + W_hr h + b_hr)   z = s(gi_z + W_hz h + b_hz)
//   n = tanh(gi_n + r * (W_hn h + b_hn))     h' = n + z * (h - n)
#include "xps_common.h"

namespace {

constexpr int GBM = 16;   // trials per workgroup

struct GruFwdParams {
    const float* gi;
    const float* w_hh[2];
    const float* b_hh[2];
    const float* h0;
    float* y_ext;
    float* saved;
    int T, B, H, ndir, Hp, ldh;
};

template <bool VEC>
__device__ inline float4 load_w4(const float* __restrict__ row, int k, int H) {
    if (VEC) {
        if (k + 3 < H) return *reinterpret_cast<const float4*>(row + k);
    }
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k + 0 < H) t.x = row[k + 0];
    if (k + 1 < H) t.y = row[k + 1];
    if (k + 2 < H) t.z = row[k + 2];
    if (k + 3 < H) t.w = row[k + 3];
    return t;
}

template <bool VEC>
__global__ __launch_bounds__(256) void gru_fwd_kernel(GruFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int dir = blockIdx.y, b0 = blockIdx.x * GBM;
    const int T = p.T, B = p.B, H = p.H, Hp = p.Hp, ldh = p.ldh;
    const int ldy = p.ndir * H;
    const float* __restrict__ W = p.w_hh[dir];
    const float* __restrict__ bh = p.b_hh[dir];
    const float* __restrict__ gi = p.gi + (long long)dir * T * B * 3 * H;

    // hidden tile <- h0 (or zeros); pads zero in both buffers
    for (int i = tid; i < 2 * GBM * ldh; i += 256) lds[i] = 0.f;
    __syncthreads();
    {
        const int slot_h0 = (dir == 0) ? 0 : T + 1, slot_other = (dir == 0) ? T + 1 : 0;
        for (int i = tid; i < GBM * H; i += 256) {
            const int r = i / H, k = i % H, b = b0 + r;
            if (b < B) {
                float v = p.h0 ? p.h0[((long long)dir * B + b) * H + k] : 0.f;
                lds[r * ldh + k] = v;
                p.y_ext[((long long)slot_h0 * B + b) * ldy + dir * H + k] = v;
                p.y_ext[((long long)slot_other * B + b) * ldy + dir * H + k] = 0.f;
            }
        }
    }
    __syncthreads();

    const int ntile = Hp / 16;
    for (int s = 0; s < T; ++s) {
        const int t = (dir == 0) ? s : T - 1 - s;
        const float* hc = lds + (s & 1) * GBM * ldh;
        float* hn = lds + ((s & 1) ^ 1) * GBM * ldh;

        for (int jt = wave; jt < ntile; jt += 4) {
            const int j = jt * 16 + n;
            const int jc = j < H ? j : H - 1;
            // input projections and biases for the epilogue, issued before the k loop
            float g_r[4], g_z[4], g_n[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int b = b0 + 4 * kq + i;
                b = b < B ? b : B - 1;
                const float* gp = gi + ((long long)t * B + b) * 3 * H;
                g_r[i] = gp[jc];
                g_z[i] = gp[H + jc];
                g_n[i] = gp[2 * H + jc];
            }
            const float bias_r = bh[jc], bias_z = bh[H + jc], bias_n = bh[2 * H + jc];

            const float* wr = W + (long long)jc * H;
            const float* wz = W + (long long)(H + jc) * H;
            const float* wn = W + (long long)(2 * H + jc) * H;
            f32x4 acc_r = {0.f, 0.f, 0.f, 0.f}, acc_z = acc_r, acc_n = acc_r;

            float4 br = load_w4<VEC>(wr, 4 * kq, H);
            float4 bz = load_w4<VEC>(wz, 4 * kq, H);
            float4 bn = load_w4<VEC>(wn, 4 * kq, H);
            for (int k0 = 0; k0 < Hp; k0 += 16) {
                const int kk = k0 + 4 * kq;
                const float4 a = *reinterpret_cast<const float4*>(hc + n * ldh + kk);
                const float4 cr = br, cz = bz, cn = bn;
                if (k0 + 16 < Hp) {
                    br = load_w4<VEC>(wr, kk + 16, H);
                    bz = load_w4<VEC>(wz, kk + 16, H);
                    bn = load_w4<VEC>(wn, kk + 16, H);
                }
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, cr.x, acc_r, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, cz.x, acc_z, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, cn.x, acc_n, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, cr.y, acc_r, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, cz.y, acc_z, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, cn.y, acc_n, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, cr.z, acc_r, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, cz.z, acc_z, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, cn.z, acc_n, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, cr.w, acc_r, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, cz.w, acc_z, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, cn.w, acc_n, 0, 0, 0);
            }

            // gates + hidden update; D layout: row = 4*kq + i, col = n
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * kq + i, b = b0 + r;
                float hnew = 0.f;
                if (b < B && j < H) {
                    const float rg = sigmoidf_acc(g_r[i] + acc_r[i] + bias_r);
                    const float zg = sigmoidf_acc(g_z[i] + acc_z[i] + bias_z);
                    const float q = acc_n[i] + bias_n;
                    const float ng = tanhf(g_n[i] + rg * q);
                    const float hp = hc[r * ldh + j];
                    hnew = ng + zg * (hp - ng);
                    p.y_ext[((long long)(t + 1) * B + b) * ldy + dir * H + j] = hnew;
                    if (p.saved) {
                        float* sv = p.saved + (((long long)dir * T + t) * B + b) * 4 * H;
                        sv[j] = rg;
                        sv[H + j] = zg;
                        sv[2 * H + j] = ng;
                        sv[3 * H + j] = q;
                    }
                }
                hn[r * ldh + j] = hnew;
            }
        }
        __syncthreads();
    }
}

struct GruBwdParams {
    const float* dy;
    const float* y_ext;
    const float* saved;
    const float* w_hh_t[2];
    float* dgi;
    float* dgh;
    float* dh0;
    int T, B, H, ndir, Hp, ldg, ldc;
};

// Backward through time.  Per step: (1) lane-parallel gate gradients from the saved
// activations -> dgi / dgh to HBM and the dgh tile (the MFMA A operand, K = 3H) to
// LDS; (2) dh_{t-1} = z * dh_t + dgh W_hh on the MFMA, W_hh^T rows streamed as
// 16-byte vectors; the running dh tile stays in LDS across steps.
template <bool VEC>
__global__ __launch_bounds__(256) void gru_bwd_kernel(GruBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int dir = blockIdx.y, b0 = blockIdx.x * GBM;
    const int T = p.T, B = p.B, H = p.H, Hp = p.Hp, ldg = p.ldg, ldc = p.ldc;
    const int ldy = p.ndir * H;
    float* G = lds;                    // [16][ldg]   dgh tile, column g*Hp + j
    float* Cy = lds + GBM * ldg;       // [16][ldc]   running dh
    const float* __restrict__ WT = p.w_hh_t[dir];   // (H x 3H)

    for (int i = tid; i < GBM * (ldg + ldc); i += 256) lds[i] = 0.f;
    __syncthreads();

    const int ntile = Hp / 16;
    for (int s = T - 1; s >= 0; --s) {
        const int t = (dir == 0) ? s : T - 1 - s;
        const int slot_prev = (dir == 0) ? t : t + 2;
        for (int idx = tid; idx < GBM * Hp; idx += 256) {
            const int r = idx / Hp, j = idx % Hp, b = b0 + r;
            float dar = 0.f, daz = 0.f, danr = 0.f, keep = 0.f;
            if (b < B && j < H) {
                const float dh = p.dy[((long long)t * B + b) * ldy + dir * H + j] + Cy[r * ldc + j];
                const float* sv = p.saved + (((long long)dir * T + t) * B + b) * 4 * H;
                const float rg = sv[j], zg = sv[H + j], ng = sv[2 * H + j], q = sv[3 * H + j];
                const float hp = p.y_ext[((long long)slot_prev * B + b) * ldy + dir * H + j];
                const float dn = dh * (1.f - zg);
                const float dz = dh * (hp - ng);
                const float dan = dn * (1.f - ng * ng);
                daz = dz * zg * (1.f - zg);
                dar = dan * q * rg * (1.f - rg);
                danr = dan * rg;
                keep = dh * zg;
                const long long o = (((long long)dir * T + t) * B + b) * 3 * H;
                p.dgi[o + j] = dar;
                p.dgi[o + H + j] = daz;
                p.dgi[o + 2 * H + j] = dan;
                p.dgh[o + j] = dar;
                p.dgh[o + H + j] = daz;
                p.dgh[o + 2 * H + j] = danr;
            }
            G[r * ldg + j] = dar;
            G[r * ldg + Hp + j] = daz;
            G[r * ldg + 2 * Hp + j] = danr;
            Cy[r * ldc + j] = keep;
        }
        __syncthreads();

        for (int jt = wave; jt < ntile; jt += 4) {
            const int j = jt * 16 + n;
            const int jc = j < H ? j : H - 1;
            f32x4 acc;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = Cy[(4 * kq + i) * ldc + j];
            const float* wrow = WT + (long long)jc * 3 * H;
#pragma unroll 1
            for (int g = 0; g < 3; ++g) {
                const float* wg = wrow + g * H;
                const float* ag = G + n * ldg + g * Hp;
                float4 bw = load_w4<VEC>(wg, 4 * kq, H);
                for (int k0 = 0; k0 < Hp; k0 += 16) {
                    const int kk = k0 + 4 * kq;
                    const float4 a = *reinterpret_cast<const float4*>(ag + kk);
                    const float4 c = bw;
                    if (k0 + 16 < Hp) bw = load_w4<VEC>(wg, kk + 16, H);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, c.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, c.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, c.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, c.w, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * kq + i;
                Cy[r * ldc + j] = (b0 + r < B && j < H) ? acc[i] : 0.f;
            }
        }
        __syncthreads();
    }

    if (p.dh0) {
        for (int i = tid; i < GBM * H; i += 256) {
            const int r = i / H, k = i % H, b = b0 + r;
            if (b < B) p.dh0[((long long)dir * B + b) * H + k] = Cy[r * ldc + k];
        }
    }
}

__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int r = by + i, c = bx + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? src[(long long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int c = bx + i, r = by + threadIdx.x;   // dst is (cols x rows)
        if (c < cols && r < rows) dst[(long long)c * rows + r] = tile[threadIdx.x][i];
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int xps_gru_seq_fwd_f32(const float* gi, const float* const* w_hh, const float* const* b_hh,
                                   const float* h0, float* y_ext, float* saved,
                                   int T, int B, int H, int ndir, void* stream) {
    XPS_CHECK_ARG(gi && w_hh && b_hh && y_ext, "null argument");
    XPS_CHECK_ARG(T >= 1 && B >= 1 && H >= 1, "T, B, H must be >= 1");
    XPS_CHECK_ARG(ndir == 1 || ndir == 2, "ndir must be 1 or 2");
    GruFwdParams p;
    p.gi = gi; p.h0 = h0; p.y_ext = y_ext; p.saved = saved;
    p.T = T; p.B = B; p.H = H; p.ndir = ndir;
    p.Hp = ((H + 15) / 16) * 16;
    p.ldh = p.Hp + 4;
    bool vec = (H % 4 == 0);
    for (int d = 0; d < 2; ++d) {
        p.w_hh[d] = w_hh[d < ndir ? d : 0];
        p.b_hh[d] = b_hh[d < ndir ? d : 0];
        XPS_CHECK_ARG(p.w_hh[d] && p.b_hh[d], "null weight pointer");
        vec = vec && aligned16(p.w_hh[d]);
    }
    const size_t lds_bytes = (size_t)2 * GBM * p.ldh * sizeof(float);
    XPS_CHECK_ARG(lds_bytes <= 160 * 1024, "hidden size too large for the LDS-resident state tile");
    dim3 grid(cdiv(B, GBM), ndir);
    if (vec) {
        if (lds_bytes > 64 * 1024)
            if (hipFuncSetAttribute((const void*)gru_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) { xps_set_error("hipFuncSetAttribute failed"); return XPS_E_HIP; }
        hipLaunchKernelGGL(gru_fwd_kernel<true>, grid, dim3(256), lds_bytes, (hipStream_t)stream, p);
    } else {
        if (lds_bytes > 64 * 1024)
            if (hipFuncSetAttribute((const void*)gru_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) { xps_set_error("hipFuncSetAttribute failed"); return XPS_E_HIP; }
        hipLaunchKernelGGL(gru_fwd_kernel<false>, grid, dim3(256), lds_bytes, (hipStream_t)stream, p);
    }
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_gru_seq_bwd_f32(const float* dy, const float* y_ext, const float* saved,
                                   const float* const* w_hh_t, float* dgi, float* dgh, float* dh0,
                                   int T, int B, int H, int ndir, void* stream) {
    XPS_CHECK_ARG(dy && y_ext && saved && w_hh_t && dgi && dgh, "null argument");
    XPS_CHECK_ARG(T >= 1 && B >= 1 && H >= 1, "T, B, H must be >= 1");
    XPS_CHECK_ARG(ndir == 1 || ndir == 2, "ndir must be 1 or 2");
    GruBwdParams p;
    p.dy = dy; p.y_ext = y_ext; p.saved = saved; p.dgi = dgi; p.dgh = dgh; p.dh0 = dh0;
    p.T = T; p.B = B; p.H = H; p.ndir = ndir;
    p.Hp = ((H + 15) / 16) * 16;
    p.ldg = 3 * p.Hp + 4;
    p.ldc = p.Hp + 4;
    bool vec = (H % 4 == 0);
    for (int d = 0; d < 2; ++d) {
        p.w_hh_t[d] = w_hh_t[d < ndir ? d : 0];
        XPS_CHECK_ARG(p.w_hh_t[d], "null weight pointer");
        vec = vec && aligned16(p.w_hh_t[d]);
    }
    const size_t lds_bytes = (size_t)GBM * (p.ldg + p.ldc) * sizeof(float);
    XPS_CHECK_ARG(lds_bytes <= 160 * 1024, "hidden size too large for the LDS-resident gradient tile");
    dim3 grid(cdiv(B, GBM), ndir);
    if (vec) {
        if (lds_bytes > 64 * 1024)
            if (hipFuncSetAttribute((const void*)gru_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) { xps_set_error("hipFuncSetAttribute failed"); return XPS_E_HIP; }
        hipLaunchKernelGGL(gru_bwd_kernel<true>, grid, dim3(256), lds_bytes, (hipStream_t)stream, p);
    } else {
        if (lds_bytes > 64 * 1024)
            if (hipFuncSetAttribute((const void*)gru_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) { xps_set_error("hipFuncSetAttribute failed"); return XPS_E_HIP; }
        hipLaunchKernelGGL(gru_bwd_kernel<false>, grid, dim3(256), lds_bytes, (hipStream_t)stream, p);
    }
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_transpose_f32(const float* src, float* dst, int rows, int cols, void* stream) {
    XPS_CHECK_ARG(src && dst && rows >= 0 && cols >= 0, "bad argument");
    if (rows == 0 || cols == 0) return XPS_OK;
    dim3 grid(cdiv(cols, 32), cdiv(rows, 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, src, dst, rows, cols);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}
