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
#include <stdlib.h>
#include "xps_common.h"
#include "xps_gemm_tile.h"
using xps_tile::bf16x4;
using xps_tile::bf16x8;
using xps_tile::bf_split;

// cluster-persistent recurrence for 128 < H <= 512 (xps_gru_cluster.hip)
bool xps_internal_gru_cluster_usable(int B, int H, int ndir);
size_t xps_internal_gru_cluster_fwd_workspace(int B, int H, int ndir);
size_t xps_internal_gru_cluster_bwd_workspace(int B, int H, int ndir);
size_t xps_internal_gru_cluster_status_offset(int B, int H, int ndir);
int xps_internal_gru_cluster_fwd(const float* gi, const float* const* w_hh, const float* const* b_hh, const float* h0,
                                 float* y_ext, float* saved, int T, int B, int H, int ndir, void* workspace, hipStream_t st,
                                 float* y_split, float* yd_split, float drop_p, unsigned long long drop_seed);
int xps_internal_gru_cluster_bwd(const float* dy, const float* dhn, const float* y_ext, const float* saved,
                                 const float* const* w_hh_t, float* dgi, float* dghn, float* dh0,
                                 int T, int B, int H, int ndir, void* workspace, hipStream_t st, int split_out);
bool xps_internal_gru_cluster_ximg_ok(int B, int H, int ndir);

namespace {

constexpr int GBM = 16;   // trials per workgroup
#ifndef XPS_GRU_BF_WAVES
#define XPS_GRU_BF_WAVES 8
#endif
constexpr int GRU_BF_WAVES = XPS_GRU_BF_WAVES;   // waves per workgroup of the H = 128 bf16 recurrence kernels

struct GruFwdParams {
    const float* gi;
    const float* w_hh[2];
    const float* b_hh[2];
    const float* h0;
    float* y_ext;
    float* saved;
    int T, B, H, ndir, Hp, ldh;
    // fused inter-layer dropout (resident kernels only): y_drop (T x B x ndir*H) = y * keep * scale, keep = dropout_keep4(seed, index / 4)
    float* y_drop;
    float drop_p, drop_scale;
    unsigned long long drop_seed;
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

__device__ inline float f4get(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

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

            // W_hh streams from L2: keep TWO groups of PF k-chunks (3 gates x PF x 16 B per lane) in flight so
            // that the memory-level parallelism, not the L2 latency, sets the rate
            constexpr int PF = 4;
            float4 wb[2][PF][3];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int kk = 16 * u + 4 * kq;
                wb[0][u][0] = load_w4<VEC>(wr, kk, H);
                wb[0][u][1] = load_w4<VEC>(wz, kk, H);
                wb[0][u][2] = load_w4<VEC>(wn, kk, H);
            }
            for (int k0 = 0; k0 < Hp; k0 += 32 * PF) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int kb = k0 + half * 16 * PF;            // first k of the group being multiplied
                    if (kb < Hp) {
#pragma unroll
                        for (int u = 0; u < PF; ++u) {             // request the following group
                            const int kk = kb + 16 * PF + 16 * u + 4 * kq;
                            wb[half ^ 1][u][0] = load_w4<VEC>(wr, kk, H);
                            wb[half ^ 1][u][1] = load_w4<VEC>(wz, kk, H);
                            wb[half ^ 1][u][2] = load_w4<VEC>(wn, kk, H);
                        }
#pragma unroll
                        for (int u = 0; u < PF; ++u) {
                            const int kc = kb + 16 * u;
                            if (kc < Hp) {
                                const float4 a = *reinterpret_cast<const float4*>(hc + n * ldh + kc + 4 * kq);
                                const float4 cr = wb[half][u][0], cz = wb[half][u][1], cn = wb[half][u][2];
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
                        }
                    }
                }
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
    const float* dhn;       // (ndir x B x H) gradient w.r.t. the FINAL hidden state of each direction, or null
    const float* dy;        // (T x B x ndir*H) gradient w.r.t. every step's output, or null
    const float* y_ext;
    const float* saved;
    const float* w_hh_t[2];
    float* dgi;
    float* dghn;
    float* dh0;
    int T, B, H, ndir, Hp, ldg, ldc;
    // dy is the gradient w.r.t. the DROPPED output of the forward kernel (resident kernels only): dy * keep * scale on the fly
    int has_drop;
    float drop_p, drop_scale;
    unsigned long long drop_seed;
    // != 0 (resident bf16x3 kernels only): dgi / dghn are written as XPS_FMT_SPLIT4 groups (xps.h) for the GEMMs that read them
    int split_out;
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
    if (p.dhn) {
        for (int i = tid; i < GBM * H; i += 256) {
            const int r = i / H, k = i % H, b = b0 + r;
            if (b < B) Cy[r * ldc + k] = p.dhn[((long long)dir * B + b) * H + k];
        }
        __syncthreads();
    }

    const int ntile = Hp / 16;
    for (int s = T - 1; s >= 0; --s) {
        const int t = (dir == 0) ? s : T - 1 - s;
        const int slot_prev = (dir == 0) ? t : t + 2;
        if (VEC) {
            // 4 consecutive hidden units per thread, 16-byte accesses (H % 4 == 0 on this path; Hp - H < 16 pad
            // columns are written as zeros)
            const int Hq = Hp / 4;
            for (int idx = tid; idx < GBM * Hq; idx += 256) {
                const int r = idx / Hq, j = (idx % Hq) * 4, b = b0 + r;
                float4 dar = make_float4(0.f, 0.f, 0.f, 0.f), daz = dar, danr = dar, keep = dar;
                if (b < B && j < H) {
                    const float4 dy4 = p.dy ? *reinterpret_cast<const float4*>(p.dy + ((long long)t * B + b) * ldy + dir * H + j)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 cy4 = *reinterpret_cast<const float4*>(&Cy[r * ldc + j]);
                    const float* sv = p.saved + (((long long)dir * T + t) * B + b) * 4 * H;
                    const float4 rg = *reinterpret_cast<const float4*>(sv + j);
                    const float4 zg = *reinterpret_cast<const float4*>(sv + H + j);
                    const float4 ng = *reinterpret_cast<const float4*>(sv + 2 * H + j);
                    const float4 q = *reinterpret_cast<const float4*>(sv + 3 * H + j);
                    const float4 hp = *reinterpret_cast<const float4*>(p.y_ext + ((long long)slot_prev * B + b) * ldy + dir * H + j);
                    float o_dar[4], o_daz[4], o_dan[4], o_danr[4], o_keep[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float dh = f4get(dy4, i) + f4get(cy4, i);
                        const float r_ = f4get(rg, i), z_ = f4get(zg, i), n_ = f4get(ng, i);
                        const float dn = dh * (1.f - z_);
                        const float dz = dh * (f4get(hp, i) - n_);
                        const float dan = dn * (1.f - n_ * n_);
                        o_daz[i] = dz * z_ * (1.f - z_);
                        o_dar[i] = dan * f4get(q, i) * r_ * (1.f - r_);
                        o_dan[i] = dan;
                        o_danr[i] = dan * r_;
                        o_keep[i] = dh * z_;
                    }
                    dar = make_float4(o_dar[0], o_dar[1], o_dar[2], o_dar[3]);
                    daz = make_float4(o_daz[0], o_daz[1], o_daz[2], o_daz[3]);
                    danr = make_float4(o_danr[0], o_danr[1], o_danr[2], o_danr[3]);
                    keep = make_float4(o_keep[0], o_keep[1], o_keep[2], o_keep[3]);
                    const long long o = (((long long)dir * T + t) * B + b) * 3 * H;
                    *reinterpret_cast<float4*>(p.dgi + o + j) = dar;
                    *reinterpret_cast<float4*>(p.dgi + o + H + j) = daz;
                    *reinterpret_cast<float4*>(p.dgi + o + 2 * H + j) = make_float4(o_dan[0], o_dan[1], o_dan[2], o_dan[3]);
                    *reinterpret_cast<float4*>(p.dghn + (((long long)dir * T + t) * B + b) * H + j) = danr;
                }
                *reinterpret_cast<float4*>(&G[r * ldg + j]) = dar;
                *reinterpret_cast<float4*>(&G[r * ldg + Hp + j]) = daz;
                *reinterpret_cast<float4*>(&G[r * ldg + 2 * Hp + j]) = danr;
                *reinterpret_cast<float4*>(&Cy[r * ldc + j]) = keep;
            }
        } else
        for (int idx = tid; idx < GBM * Hp; idx += 256) {
            const int r = idx / Hp, j = idx % Hp, b = b0 + r;
            float dar = 0.f, daz = 0.f, danr = 0.f, keep = 0.f;
            if (b < B && j < H) {
                const float dh = (p.dy ? p.dy[((long long)t * B + b) * ldy + dir * H + j] : 0.f) + Cy[r * ldc + j];
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
                p.dghn[(((long long)dir * T + t) * B + b) * H + j] = danr;
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
            f32x4 acc, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = Cy[(4 * kq + i) * ldc + j];
            const float* wrow = WT + (long long)jc * 3 * H;
#pragma unroll 1
            for (int g = 0; g < 3; ++g) {
                const float* wg = wrow + g * H;
                const float* ag = G + n * ldg + g * Hp;
                constexpr int PF = 8;                       // k-chunks per prefetch group (one gate segment at a time)
                float4 wb[2][PF];
#pragma unroll
                for (int u = 0; u < PF; ++u) wb[0][u] = load_w4<VEC>(wg, 16 * u + 4 * kq, H);
                for (int k0 = 0; k0 < Hp; k0 += 32 * PF) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int kb = k0 + half * 16 * PF;
                        if (kb < Hp) {
#pragma unroll
                            for (int u = 0; u < PF; ++u)
                                wb[half ^ 1][u] = load_w4<VEC>(wg, kb + 16 * PF + 16 * u + 4 * kq, H);
#pragma unroll
                            for (int u = 0; u < PF; ++u) {
                                const int kc = kb + 16 * u;
                                if (kc < Hp) {
                                    const float4 a = *reinterpret_cast<const float4*>(ag + kc + 4 * kq);
                                    const float4 c = wb[half][u];
                                    // two accumulator chains (16x16x4: 32-cycle issue, 40-cycle dependent latency)
                                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, c.x, acc, 0, 0, 0);
                                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, c.y, acc2, 0, 0, 0);
                                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, c.z, acc, 0, 0, 0);
                                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, c.w, acc2, 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * kq + i;
                Cy[r * ldc + j] = (b0 + r < B && j < H) ? acc[i] + acc2[i] : 0.f;
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


// ---------------------------------------------------------------------------------------------
// Register-resident specialisation (H = 64 or 128).  W_hh (3H x H fp32; 196 KB at H = 128) is
// loaded ONCE into the VGPRs of the workgroup's four waves (wave w owns hidden tiles w and w+4:
// 2 tiles x 3 gates x 8 k-chunks x float4 = 192 VGPRs; one wave per SIMD, 512-register budget)
// and reused for every time step: per step a wave issues 192 back-to-back MFMAs fed by 8
// ds_read_b128 of the LDS hidden tile — no global or L2 traffic for weights inside the loop.
//
// Operand roles are SWAPPED relative to the streaming kernel: A = W_hh fragment (row = hidden
// unit), B = h^T (col = trial), so D[row = unit 4*kq + i][col = trial n]: every lane owns FOUR
// CONSECUTIVE hidden units of ONE trial, and all per-step traffic (gi in, h out, saved gates,
// LDS state) is 16-byte vectors.  Activations use v_exp_f32 / v_rcp_f32 (abs error ~1e-7,
// far inside the 1e-4 logit budget).
// ---------------------------------------------------------------------------------------------
__device__ inline float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ inline float fast_tanh(float x) { return 2.0f * fast_sigmoid(2.0f * x) - 1.0f; }

#ifdef XPS_STAMP
// Diagnostic build only (never shipped): per-wave cycle shares of the step loop, written to a
// buffer of their own that no kernel reads.
__device__ unsigned long long g_stamp[4096 * 8];
#define STAMP(var)                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);
#else
#define STAMP(var)
#endif

// BF: the recurrent product h W_hh^T runs on the bf16 matrix pipe with split operands (W_hh split once into hi/lo
// register fragments, h split when a step writes it to LDS; three v_mfma_f32_16x16x32_bf16 per product, see
// xps_gemm_tile.h): the MFMA phase of a step shrinks from 192 x 32 to 72 x 16 cycles.
// NW waves per workgroup (4 or 8): with 8, every SIMD holds two waves of the workgroup (each owning one 16-unit tile and half
// the W_hh fragments), which hides the LDS / MFMA-result / barrier latencies that a single wave per SIMD exposes.
template <int H, bool BF = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, 1) void gru_fwd_resident_kernel(GruFwdParams p) {
    constexpr int NTHR = NW * 64;
    // LDS row strides of 8 dwords (mod 64): the b128 fragment reads (16 trials x 4 k-quarters; lane groups {0-3,12-15,20-27}, ...:
    // MI355X_MICROARCH.md, LDS) then touch 16 distinct 4-bank groups per service cycle.  (4 (mod 64), the round-1 value, put
    // trials 11 / kq 1 and 12 / kq 0 on the same banks: 40 % of the LDS-active cycles were conflicts.)
#ifdef XPS_GRU_OLD_PAD
    constexpr int NT = H / 16, TPW = NT / NW, NC = H / 16, LDH = H + 4;
#else
    constexpr int NT = H / 16, TPW = NT / NW, NC = H / 16, LDH = H + 8;
#endif
    static_assert(NT % NW == 0, "hidden tiles must divide among the waves");
#ifdef XPS_GRU_OLD_PAD
    constexpr int NCB = H / 32, LDB = H + 8;
#else
    constexpr int NCB = H / 32, LDB = H + 16;  // BF: 32-wide k chunks; bf16 row stride (conflict-free b128 reads)
#endif
    constexpr int NST = TPW * 6;               // 16-byte stores per lane and step: h, r, z, n, q (, dropped h) per tile
    __shared__ __attribute__((aligned(16))) float hs[2][GBM][LDH];
    __shared__ __attribute__((aligned(16))) __bf16 hsb[BF ? 2 : 1][2][BF ? GBM : 1][BF ? LDB : 8];   // [buffer][hi, lo][trial][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int dir = blockIdx.y, b0 = blockIdx.x * GBM;
    const int T = p.T, B = p.B;
    const int ldy = p.ndir * H;
    const float* __restrict__ W = p.w_hh[dir];
    const float* __restrict__ bh = p.b_hh[dir];
    const float* __restrict__ gi = p.gi + (long long)dir * T * B * 3 * H;
    const int b = b0 + n;                      // this lane's trial
    const bool live = b < B;
    const int bc = live ? b : B - 1;
    const bool do_save = p.saved != nullptr;
#ifdef XPS_STAMP
    unsigned long long st_entry = 0, st_loop = 0, st_exit = 0;
    STAMP(st_entry)
#endif

    float w[BF ? 1 : TPW][3][NC][4];           // A operand: W[g*H + j0 + n][16c + 4kq + e]
    bf16x8 wh[BF ? TPW : 1][3][NCB], wl[BF ? TPW : 1][3][NCB];   // BF: W[g*H + j0 + n][32c + 8kq + j] split hi / lo
    float4 bias[TPW][3];                       // b_hh of the lane's 4 output units
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int j0 = (wave + NW * tt) * 16;
            bias[tt][g] = *reinterpret_cast<const float4*>(bh + g * H + j0 + 4 * kq);
            if constexpr (BF) {
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    const float* wp = W + (long long)(g * H + j0 + n) * H + 32 * c + 8 * kq;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(wp), v1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        __bf16 a, b;
                        bf_split(v0[j], a, b); wh[tt][g][c][j] = a; wl[tt][g][c][j] = b;
                        bf_split(v1[j], a, b); wh[tt][g][c][4 + j] = a; wl[tt][g][c][4 + j] = b;
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float4 v = *reinterpret_cast<const float4*>(W + (long long)(g * H + j0 + n) * H + 16 * c + 4 * kq);
                    w[tt][g][c][0] = v.x; w[tt][g][c][1] = v.y; w[tt][g][c][2] = v.z; w[tt][g][c][3] = v.w;
                }
            }
        }

    for (int i = tid; i < 2 * GBM * LDH; i += NTHR) (&hs[0][0][0])[i] = 0.f;
    __syncthreads();
    {
        const int slot_h0 = (dir == 0) ? 0 : T + 1, slot_other = (dir == 0) ? T + 1 : 0;
        for (int i = tid; i < GBM * H; i += NTHR) {
            const int r = i / H, k = i % H, bb = b0 + r;
            if (bb < B) {
                const float v = p.h0 ? p.h0[((long long)dir * B + bb) * H + k] : 0.f;
                hs[0][r][k] = v;
                p.y_ext[((long long)slot_h0 * B + bb) * ldy + dir * H + k] = v;
                p.y_ext[((long long)slot_other * B + bb) * ldy + dir * H + k] = 0.f;
            }
        }
    }
    __syncthreads();
    if constexpr (BF) {
        for (int i = tid; i < GBM * H; i += NTHR) {
            const int r = i / H, k = i % H;
            __bf16 a, b;
            bf_split(hs[0][r][k], a, b);
            hsb[0][0][r][k] = a; hsb[0][1][r][k] = b;
            hsb[1][0][r][k] = (__bf16)0.f; hsb[1][1][r][k] = (__bf16)0.f;
        }
        __syncthreads();
    }

    // Results of step s are NOT stored at the end of step s: a CU retires stores at ~16 B/clk, so 40 KB
    // per step would stall every wave ~2400 cycles at issue.  They are kept in registers (pend) and
    // issued one per k-chunk inside the MFMA loop of step s+1, where the VMEM pipe is otherwise idle.
    float4 pend[NST];
    float* pend_y = nullptr;
    float* pend_sv = nullptr;
    float* pend_yd = nullptr;
    const bool do_drop = p.y_drop != nullptr;
    auto issue_store = [&](int k) {            // k is a compile-time constant at every call site
        const int tt = k % TPW, what = k / TPW;
        const int j = (wave + NW * tt) * 16 + 4 * kq;
        if (what == 0) *reinterpret_cast<float4*>(pend_y + j) = pend[k];
        else if (what == 5) { if (do_drop) *reinterpret_cast<float4*>(pend_yd + j) = pend[k]; }
        else if (do_save) *reinterpret_cast<float4*>(pend_sv + (what - 1) * H + j) = pend[k];
    };

    // gi of step s+1 is requested before step s's epilogue so its latency hides under the gate math
    float4 g_nxt[TPW][3];
    {
        const int t0 = (dir == 0) ? 0 : T - 1;
        const float* gp = gi + ((long long)t0 * B + bc) * 3 * H;
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                g_nxt[tt][g] = *reinterpret_cast<const float4*>(gp + g * H + (wave + NW * tt) * 16 + 4 * kq);
    }
#ifdef XPS_STAMP
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, acc_mfma = 0, acc_epi = 0, acc_store = 0, acc_bar = 0;
    STAMP(st_loop)
#endif
    for (int s = 0; s < T; ++s) {
        const int t = (dir == 0) ? s : T - 1 - s;
        const int cur = s & 1;
        STAMP(st0)
        float4 g_in[TPW][3];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
            for (int g = 0; g < 3; ++g) g_in[tt][g] = g_nxt[tt][g];

        f32x4 acc[TPW][3];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[tt][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool flush = (s > 0) && live;
        if constexpr (BF) {
            bf16x8 hbh[2], hbl[2];
            hbh[0] = *reinterpret_cast<const bf16x8*>(&hsb[cur][0][n][8 * kq]);
            hbl[0] = *reinterpret_cast<const bf16x8*>(&hsb[cur][1][n][8 * kq]);
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                if (c + 1 < NCB) {
                    hbh[(c + 1) & 1] = *reinterpret_cast<const bf16x8*>(&hsb[cur][0][n][32 * (c + 1) + 8 * kq]);
                    hbl[(c + 1) & 1] = *reinterpret_cast<const bf16x8*>(&hsb[cur][1][n][32 * (c + 1) + 8 * kq]);
                }
                const bf16x8 bh8 = hbh[c & 1], bl8 = hbl[c & 1];
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        acc[tt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[tt][g][c], bh8, acc[tt][g], 0, 0, 0);
                        acc[tt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[tt][g][c], bl8, acc[tt][g], 0, 0, 0);
                        acc[tt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[tt][g][c], bh8, acc[tt][g], 0, 0, 0);
                    }
                if (flush) {
#pragma unroll
                    for (int k = c; k < NST; k += NCB) issue_store(k);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        float4 hb[2];
        hb[0] = *reinterpret_cast<const float4*>(&hs[cur][n][4 * kq]);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c + 1 < NC) hb[(c + 1) & 1] = *reinterpret_cast<const float4*>(&hs[cur][n][16 * (c + 1) + 4 * kq]);
            const float4 a4 = hb[c & 1];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc[tt][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tt][g][c][e], f4get(a4, e), acc[tt][g], 0, 0, 0);
            if (flush) {
#pragma unroll
                for (int k = c; k < NST; k += NC) issue_store(k);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        STAMP(st1)
        if (s + 1 < T) {
            const int tn = (dir == 0) ? s + 1 : T - 2 - s;
            const float* gp = gi + ((long long)tn * B + bc) * 3 * H;
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    g_nxt[tt][g] = *reinterpret_cast<const float4*>(gp + g * H + (wave + NW * tt) * 16 + 4 * kq);
        }
        // gates: lane owns units j0 + 4*kq + {0..3} of trial b
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int j = (wave + NW * tt) * 16 + 4 * kq;
            const float4 hp = *reinterpret_cast<const float4*>(&hs[cur][n][j]);
            float o[4], r_[4], z_[4], n_[4], q_[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float rg = fast_sigmoid(f4get(g_in[tt][0], i) + acc[tt][0][i] + f4get(bias[tt][0], i));
                const float zg = fast_sigmoid(f4get(g_in[tt][1], i) + acc[tt][1][i] + f4get(bias[tt][1], i));
                const float q = acc[tt][2][i] + f4get(bias[tt][2], i);
                const float ng = fast_tanh(f4get(g_in[tt][2], i) + rg * q);
                o[i] = live ? ng + zg * (f4get(hp, i) - ng) : 0.f;
                r_[i] = rg; z_[i] = zg; n_[i] = ng; q_[i] = q;
            }
            pend[0 * TPW + tt] = make_float4(o[0], o[1], o[2], o[3]);
            pend[1 * TPW + tt] = make_float4(r_[0], r_[1], r_[2], r_[3]);
            pend[2 * TPW + tt] = make_float4(z_[0], z_[1], z_[2], z_[3]);
            pend[3 * TPW + tt] = make_float4(n_[0], n_[1], n_[2], n_[3]);
            pend[4 * TPW + tt] = make_float4(q_[0], q_[1], q_[2], q_[3]);
            if (do_drop) {
                // element index inside the (T, B, ndir * H) output = the index xps_dropout_f32 would see
                const f32x4 m = dropout_keep4(p.drop_seed, (((long long)t * B + bc) * ldy + dir * H + j) >> 2, p.drop_p);
                pend[5 * TPW + tt] = make_float4(o[0] * m[0] * p.drop_scale, o[1] * m[1] * p.drop_scale, o[2] * m[2] * p.drop_scale,
                                                 o[3] * m[3] * p.drop_scale);
            }
            *reinterpret_cast<float4*>(&hs[cur ^ 1][n][j]) = pend[tt];
            if constexpr (BF) {
                bf16x4 sh, sl;
#pragma unroll
                for (int i = 0; i < 4; ++i) { __bf16 a, b; bf_split(o[i], a, b); sh[i] = a; sl[i] = b; }
                *reinterpret_cast<bf16x4*>(&hsb[cur ^ 1][0][n][j]) = sh;
                *reinterpret_cast<bf16x4*>(&hsb[cur ^ 1][1][n][j]) = sl;
            }
        }
        pend_y = p.y_ext + ((long long)(t + 1) * B + bc) * ldy + dir * H;
        pend_sv = do_save ? p.saved + (((long long)dir * T + t) * B + bc) * 4 * H : nullptr;
        pend_yd = do_drop ? p.y_drop + ((long long)t * B + bc) * ldy + dir * H : nullptr;
        STAMP(st2)
        STAMP(st3)
        __syncthreads();
#ifdef XPS_STAMP
        {
            unsigned long long st4;
            STAMP(st4)
            acc_mfma += st1 - st0; acc_epi += st2 - st1; acc_store += st3 - st2; acc_bar += st4 - st3;
        }
#endif
    }
    if (live) {
#pragma unroll
        for (int k = 0; k < NST; ++k) issue_store(k);
    }
#ifdef XPS_STAMP
    STAMP(st_exit)
    if (lane == 0) {
        const int wid = ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) & 4095;
        g_stamp[wid * 8 + 0] = acc_mfma; g_stamp[wid * 8 + 1] = acc_epi;
        g_stamp[wid * 8 + 2] = acc_store; g_stamp[wid * 8 + 3] = acc_bar;
        g_stamp[wid * 8 + 4] = st_loop - st_entry; g_stamp[wid * 8 + 5] = st_exit - st_loop;
        g_stamp[wid * 8 + 6] = st_entry; g_stamp[wid * 8 + 7] = st_exit;
    }
#endif
}

template <int H, bool BF = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, 1) void gru_bwd_resident_kernel(GruBwdParams p) {
    constexpr int NTHR = NW * 64;
#ifdef XPS_GRU_OLD_PAD
    constexpr int NT = H / 16, TPW = NT / NW, NC = 3 * H / 16, LDG = 3 * H + 4, LDC = H + 4;
#else
    constexpr int NT = H / 16, TPW = NT / NW, NC = 3 * H / 16, LDG = 3 * H + 8, LDC = H + 8;   // strides: 8 dwords (mod 64), see the forward kernel
#endif
    static_assert(NT % NW == 0, "hidden tiles must divide among the waves");
#ifdef XPS_GRU_OLD_PAD
    constexpr int NCB = 3 * H / 32, LDGB = 3 * H + 8;
#else
    constexpr int NCB = 3 * H / 32, LDGB = 3 * H + 16;    // BF: 32-wide k chunks; bf16 row stride of the gate gradients
#endif
    constexpr int H4 = H / 4, GPT = GBM * H4 / NTHR;       // float4 groups of the 16 x H tile per thread
    __shared__ __attribute__((aligned(16))) float G[BF ? 1 : GBM][BF ? 4 : LDG];
    __shared__ __attribute__((aligned(16))) __bf16 Gb[2][BF ? GBM : 1][BF ? LDGB : 8];      // BF: [hi, lo][trial][k]
    __shared__ __attribute__((aligned(16))) float Cy[GBM][LDC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int dir = blockIdx.y, b0 = blockIdx.x * GBM;
    const int T = p.T, B = p.B;
    const int ldy = p.ndir * H;
    const float* __restrict__ WT = p.w_hh_t[dir];   // (H x 3H)

    float w[BF ? 1 : TPW][NC][4];                    // A operand: WT[j0 + n][16c + 4kq + e]
    bf16x8 wh[BF ? TPW : 1][NCB], wl[BF ? TPW : 1][NCB];   // BF: WT[j0 + n][32c + 8kq + j] split hi / lo
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int j = (wave + NW * tt) * 16 + n;
        if constexpr (BF) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const float* wp = WT + (long long)j * 3 * H + 32 * c + 8 * kq;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(wp), v1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __bf16 a, b;
                    bf_split(v0[e], a, b); wh[tt][c][e] = a; wl[tt][c][e] = b;
                    bf_split(v1[e], a, b); wh[tt][c][4 + e] = a; wl[tt][c][4 + e] = b;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(WT + (long long)j * 3 * H + 16 * c + 4 * kq);
                w[tt][c][0] = v.x; w[tt][c][1] = v.y; w[tt][c][2] = v.z; w[tt][c][3] = v.w;
            }
        }
    }
    if constexpr (BF) {
        for (int i = tid; i < 2 * GBM * LDGB; i += NTHR) (&Gb[0][0][0])[i] = (__bf16)0.f;
    } else {
        for (int i = tid; i < GBM * LDG; i += NTHR) (&G[0][0])[i] = 0.f;
    }
    for (int i = tid; i < GBM * LDC; i += NTHR) (&Cy[0][0])[i] = 0.f;
    __syncthreads();
    if (p.dhn) {
        for (int i = tid; i < GBM * H; i += NTHR) {
            const int r = i / H, k = i % H, b = b0 + r;
            if (b < B) Cy[r][k] = p.dhn[((long long)dir * B + b) * H + k];
        }
        __syncthreads();
    }

    // inputs of the gate-gradient phase for one step: saved r,z,n,q, h_prev and dy (7 x float4 per group)
    struct StepIn { float4 dy, rg, zg, ng, q, hp; };
    StepIn nxt[GPT];
    auto load_step = [&](int s_, StepIn (&dst)[GPT]) {
        const int t_ = (dir == 0) ? s_ : T - 1 - s_;
        const int slot_prev_ = (dir == 0) ? t_ : t_ + 2;
#pragma unroll
        for (int e = 0; e < GPT; ++e) {
            const int idx = tid + NTHR * e;
            const int r = idx / H4, j = (idx % H4) * 4;
            int b = b0 + r;
            b = b < B ? b : B - 1;
            const float* sv = p.saved + (((long long)dir * T + t_) * B + b) * 4 * H;
            dst[e].dy = p.dy ? *reinterpret_cast<const float4*>(p.dy + ((long long)t_ * B + b) * ldy + dir * H + j)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
            dst[e].rg = *reinterpret_cast<const float4*>(sv + j);
            dst[e].zg = *reinterpret_cast<const float4*>(sv + H + j);
            dst[e].ng = *reinterpret_cast<const float4*>(sv + 2 * H + j);
            dst[e].q = *reinterpret_cast<const float4*>(sv + 3 * H + j);
            dst[e].hp = *reinterpret_cast<const float4*>(p.y_ext + ((long long)slot_prev_ * B + b) * ldy + dir * H + j);
        }
    };
    load_step(T - 1, nxt);

    constexpr int NST = GPT * 4;               // 16-byte stores per thread and step: dar, daz, dan, dan*r per group
    for (int s = T - 1; s >= 0; --s) {
        const int t = (dir == 0) ? s : T - 1 - s;
        StepIn in[GPT];
#pragma unroll
        for (int e = 0; e < GPT; ++e) in[e] = nxt[e];
        if (s > 0) load_step(s - 1, nxt);
        // (1) gate gradients, 4 consecutive hidden units per thread-group; results go to LDS now and to
        //     HBM later, one store per few k-chunks of the MFMA phase (the VMEM pipe is idle there)
        float4 pend[NST];
        bool pend_live[GPT];
#pragma unroll
        for (int e = 0; e < GPT; ++e) {
            const int idx = tid + NTHR * e;
            const int r = idx / H4, j = (idx % H4) * 4, b = b0 + r;
            float4 dar = make_float4(0.f, 0.f, 0.f, 0.f), daz = dar, danr = dar, keep = dar, dan4 = dar;
            pend_live[e] = b < B;
            if (b < B) {
                float4 dy4 = in[e].dy;
                if (p.has_drop) {              // dy arrives for the dropped output: the same decisions as the forward kernel's
                    const f32x4 m = dropout_keep4(p.drop_seed, (((long long)t * B + b) * ldy + dir * H + j) >> 2, p.drop_p);
                    dy4 = make_float4(dy4.x * m[0] * p.drop_scale, dy4.y * m[1] * p.drop_scale, dy4.z * m[2] * p.drop_scale,
                                      dy4.w * m[3] * p.drop_scale);
                }
                const float4 cy4 = *reinterpret_cast<const float4*>(&Cy[r][j]);
                const float4 rg = in[e].rg, zg = in[e].zg, ng = in[e].ng, q = in[e].q, hp = in[e].hp;
                float o_dar[4], o_daz[4], o_dan[4], o_danr[4], o_keep[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float dh = f4get(dy4, i) + f4get(cy4, i);
                    const float r_ = f4get(rg, i), z_ = f4get(zg, i), n_ = f4get(ng, i);
                    const float dn = dh * (1.f - z_);
                    const float dz = dh * (f4get(hp, i) - n_);
                    const float dan = dn * (1.f - n_ * n_);
                    o_daz[i] = dz * z_ * (1.f - z_);
                    o_dar[i] = dan * f4get(q, i) * r_ * (1.f - r_);
                    o_dan[i] = dan;
                    o_danr[i] = dan * r_;
                    o_keep[i] = dh * z_;
                }
                dar = make_float4(o_dar[0], o_dar[1], o_dar[2], o_dar[3]);
                daz = make_float4(o_daz[0], o_daz[1], o_daz[2], o_daz[3]);
                danr = make_float4(o_danr[0], o_danr[1], o_danr[2], o_danr[3]);
                keep = make_float4(o_keep[0], o_keep[1], o_keep[2], o_keep[3]);
                dan4 = make_float4(o_dan[0], o_dan[1], o_dan[2], o_dan[3]);
            }
            pend[e * 4 + 0] = dar; pend[e * 4 + 1] = daz; pend[e * 4 + 2] = dan4; pend[e * 4 + 3] = danr;
            if (BF && p.split_out) {        // the only readers are GEMMs: hand them the hi / lo split (same 16 bytes per group)
                auto pack = [](const float4& v) {
                    const f32x4 o = split4_pack((f32x4){v.x, v.y, v.z, v.w});
                    return make_float4(o[0], o[1], o[2], o[3]);
                };
                pend[e * 4 + 0] = pack(dar); pend[e * 4 + 1] = pack(daz); pend[e * 4 + 2] = pack(dan4); pend[e * 4 + 3] = pack(danr);
            }
            if constexpr (BF) {
                auto put = [&](const float4& v, int col) {
                    bf16x4 sh, sl;
                    __bf16 a, b;
                    bf_split(v.x, a, b); sh[0] = a; sl[0] = b;
                    bf_split(v.y, a, b); sh[1] = a; sl[1] = b;
                    bf_split(v.z, a, b); sh[2] = a; sl[2] = b;
                    bf_split(v.w, a, b); sh[3] = a; sl[3] = b;
                    *reinterpret_cast<bf16x4*>(&Gb[0][r][col]) = sh;
                    *reinterpret_cast<bf16x4*>(&Gb[1][r][col]) = sl;
                };
                put(dar, j); put(daz, H + j); put(danr, 2 * H + j);
            } else {
                *reinterpret_cast<float4*>(&G[r][j]) = dar;
                *reinterpret_cast<float4*>(&G[r][H + j]) = daz;
                *reinterpret_cast<float4*>(&G[r][2 * H + j]) = danr;
            }
            *reinterpret_cast<float4*>(&Cy[r][j]) = keep;
        }
        auto issue_store = [&](int k) {        // k compile-time at every call site
            const int e = k / 4, what = k % 4;
            const int idx = tid + NTHR * e;
            const int r = idx / H4, j = (idx % H4) * 4, b = b0 + r;
            if (pend_live[e]) {
                if (what < 3) *reinterpret_cast<float4*>(p.dgi + (((long long)dir * T + t) * B + b) * 3 * H + what * H + j) = pend[k];
                else *reinterpret_cast<float4*>(p.dghn + (((long long)dir * T + t) * B + b) * H + j) = pend[k];
            }
        };
        __syncthreads();

        // (2) dh_{t-1} = z*dh + dgh W_hh : D[row = unit 4kq+i][col = trial n]
        f32x4 acc[TPW];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const float4 c4 = *reinterpret_cast<const float4*>(&Cy[n][(wave + NW * tt) * 16 + 4 * kq]);
            acc[tt] = (f32x4){c4.x, c4.y, c4.z, c4.w};
        }
        if constexpr (BF) {
            bf16x8 gbh[2], gbl[2];
            gbh[0] = *reinterpret_cast<const bf16x8*>(&Gb[0][n][8 * kq]);
            gbl[0] = *reinterpret_cast<const bf16x8*>(&Gb[1][n][8 * kq]);
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                if (c + 1 < NCB) {
                    gbh[(c + 1) & 1] = *reinterpret_cast<const bf16x8*>(&Gb[0][n][32 * (c + 1) + 8 * kq]);
                    gbl[(c + 1) & 1] = *reinterpret_cast<const bf16x8*>(&Gb[1][n][32 * (c + 1) + 8 * kq]);
                }
                const bf16x8 bh8 = gbh[c & 1], bl8 = gbl[c & 1];
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) {
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[tt][c], bh8, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[tt][c], bl8, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[tt][c], bh8, acc[tt], 0, 0, 0);
                }
                if (c < NST) issue_store(c);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = NCB; k < NST; ++k) issue_store(k);
        } else {
        float4 gb[2];
        gb[0] = *reinterpret_cast<const float4*>(&G[n][4 * kq]);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c + 1 < NC) gb[(c + 1) & 1] = *reinterpret_cast<const float4*>(&G[n][16 * (c + 1) + 4 * kq]);
            const float4 a4 = gb[c & 1];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt)
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tt][c][e], f4get(a4, e), acc[tt], 0, 0, 0);
            if (c % 2 == 0 && c / 2 < NST) issue_store(c / 2);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = NC / 2 + (NC % 2); k < NST; ++k) issue_store(k);
        }
        const bool live = b0 + n < B;
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const float4 o = live ? make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&Cy[n][(wave + NW * tt) * 16 + 4 * kq]) = o;
        }
        __syncthreads();
    }
    if (p.dh0) {
        for (int i = tid; i < GBM * H; i += NTHR) {
            const int r = i / H, k = i % H, b = b0 + r;
            if (b < B) p.dh0[((long long)dir * B + b) * H + k] = Cy[r][k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Large hidden sizes (H > 128: W_hh no longer fits one CU's registers).  One launch PER TIME STEP;
// the kernel boundary is the inter-workgroup exchange of h_t, so no in-kernel grid synchronisation
// is needed.  Each workgroup computes, for 128 trials x 32 hidden units, the three gate
// pre-activations h_{t-1} W_hh^T on the 32x32x2 fp32 MFMA (tile machinery of xps_gemm_tile.h:
// k-major LDS staging, loads two k-tiles ahead) and applies the gate math in the epilogue: the
// r, z, n accumulators of a (trial, unit) pair live in the same lane.  B-operand rows are the
// gate-strided rows {g*H + j} of W_hh, addressed by a two-level row map.  Grid = (H/32, B/128, ndir).
// ---------------------------------------------------------------------------------------------
struct GruStepFwd {
    const float* gi;        // (ndir, T, B, 3H)
    const float* w_hh[2];
    const float* b_hh[2];
    float* y_ext;           // (T+2, B, ndir*H)
    float* saved;           // (ndir, T, B, 4H) or null
    int T, B, H, ndir, s;   // s = processing step
    int vecA, vecB;
};

template <bool BF>
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(GruStepFwd p) {
    using namespace xps_tile;
    __shared__ __attribute__((aligned(16))) TileMem<BF> mem;        // fp32 k-major tiles or the bf16 split image
    const int dir = blockIdx.z, H = p.H, B = p.B, T = p.T;
    const int t = (dir == 0) ? p.s : T - 1 - p.s;
    const int slot_prev = (dir == 0) ? t : t + 2;
    const int ldy = p.ndir * H;
    const int j0 = blockIdx.x * 32, m0 = blockIdx.y * 128;
    const float* hprev = p.y_ext + (long long)slot_prev * B * ldy + dir * H;     // (B x H), ld = ldy
    RowMap ra; ra.gs = 0; ra.ld = ldy; ra.rpg = 1 << 30;
    RowMap rb; rb.gs = (long long)H * H; rb.ld = H; rb.rpg = 32;                  // row x -> (x/32)*H*H + (x%32)*H
    const float* Wb = p.w_hh[dir] + (long long)j0 * H;
    // units beyond H in the last tile: clamp the loader's row range per gate by shrinking X to the valid rows
    const int nu = min(32, H - j0);
    f32x16 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    {
        // gemm_accumulate works on a (MI x 2) grid of 32x32 tiles per wave (64 x 64); here a wave owns
        // 32 trials x (3 gates x 32 units), so the k pipeline is restated for that shape.
        using LA = TileLoader<true, 128, BF>;
        using LB = TileLoader<true, 128, BF>;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        LA la; LB lb;
        la.init(hprev, ra, m0, B, H, tid, p.vecA);
        lb.init(Wb, rb, 0, (nu == 32) ? 96 : 0, H, tid, p.vecB);
        if (nu < 32) {                         // ragged last unit tile: valid rows are x with (x % 32) < nu
#pragma unroll
            for (int r = 0; r < LB::NV; ++r) {
                const int x = tid / KL + LB::XR * r;
                lb.xoff[r] = (x < 96 && (x & 31) < nu) ? rb.off(x) : -1;
            }
            lb.fast = false;
        }
        const int li = lane & 31, lk = lane >> 5;
        const int nkt = (H + BKT - 1) / BKT;
        f32x4 ra0[LA::NV], rb0[LB::NV], ra1[LA::NV], rb1[LB::NV];
        // the gate epilogue's inputs (input projections of this step, previous state) do not depend on the product: they
        // are requested NOW and arrive under the k loop.  Loaded row by row inside the epilogue they formed a chain of
        // sixteen dependent global round trips (the stores of row r may alias the loads of row r + 1 for the compiler).
        float e_r[16], e_z[16], e_n[16], e_h[16];
        {
            const int je = j0 + li;
            const float* gie = p.gi + ((long long)dir * T + t) * B * 3 * H;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                const bool ok = b < B && je < H;
                const float* gp = gie + (long long)(ok ? b : 0) * 3 * H + (ok ? je : 0);
                e_r[r] = ok ? gp[0] : 0.f;
                e_z[r] = ok ? gp[H] : 0.f;
                e_n[r] = ok ? gp[2 * H] : 0.f;
                e_h[r] = ok ? hprev[(long long)b * ldy + je] : 0.f;
            }
        }
        la.load(ra0, hprev, ra, m0, B, 0, H, tid, p.vecA);
        lb.load(rb0, Wb, rb, 0, 96, 0, H, tid, p.vecB);
        if (nkt > 1) {
            la.load(ra1, hprev, ra, m0, B, BKT, H, tid, p.vecA);
            lb.load(rb1, Wb, rb, 0, 96, BKT, H, tid, p.vecB);
        }
        auto stage = [&](const f32x4 (&va)[LA::NV], const f32x4 (&vb)[LB::NV], int buf) {
            if constexpr (BF) {
                bf_store<true, 128>(va, mem.st.a[buf], tid, false);
                bf_store<true, 128>(vb, mem.st.b[buf], tid, false);
            } else {
                la.store(va, mem.As[buf], tid);
                lb.store(vb, mem.Bs[buf], tid);
            }
        };
        stage(ra0, rb0, 0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            if (kt + 2 < nkt) {
                la.load(ra0, hprev, ra, m0, B, (kt + 2) * BKT, H, tid, p.vecA);
                lb.load(rb0, Wb, rb, 0, 96, (kt + 2) * BKT, H, tid, p.vecB);
            }
            if constexpr (BF) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&mem.st.a[buf].hi[wave * 32 + li][lk * 8]);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(&mem.st.a[buf].lo[wave * 32 + li][lk * 8]);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&mem.st.b[buf].hi[g * 32 + li][lk * 8]);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&mem.st.b[buf].lo[g * 32 + li][lk * 8]);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[g], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < BKT; kk += 2) {
                    const float a = mem.As[buf][kk + lk][wave * 32 + li];
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, mem.Bs[buf][kk + lk][g * 32 + li], acc[g], 0, 0, 0);
                }
            }
            if (kt + 1 < nkt) {
                stage(ra1, rb1, buf ^ 1);
#pragma unroll
                for (int r = 0; r < LA::NV; ++r) ra1[r] = ra0[r];
#pragma unroll
                for (int r = 0; r < LB::NV; ++r) rb1[r] = rb0[r];
            }
            __syncthreads();
        }
        // gates: lane (li, lk) holds unit j0 + li of trials m0 + 32*wave + (r&3) + 8*(r>>2) + 4*lk
        const int j = j0 + li;
        if (j < H) {
            const float* bh = p.b_hh[dir];
            const float b_r = bh[j], b_z = bh[H + j], b_n = bh[2 * H + j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (b < B) {
                    const float rg = sigmoidf_acc(e_r[r] + acc[0][r] + b_r);
                    const float zg = sigmoidf_acc(e_z[r] + acc[1][r] + b_z);
                    const float q = acc[2][r] + b_n;
                    const float ng = tanhf(e_n[r] + rg * q);
                    const float hp = e_h[r];
                    p.y_ext[((long long)(t + 1) * B + b) * ldy + dir * H + j] = ng + zg * (hp - ng);
                    if (p.saved) {
                        float* sv = p.saved + (((long long)dir * T + t) * B + b) * 4 * H;
                        sv[j] = rg; sv[H + j] = zg; sv[2 * H + j] = ng; sv[3 * H + j] = q;
                    }
                }
            }
        }
    }
}

// h0 slots of y_ext (slot 0 / slot T+1) for the per-step path
__global__ void gru_init_slots_kernel(const float* __restrict__ h0, float* __restrict__ y_ext, int T, int B, int H, int ndir) {
    const long long n = (long long)ndir * B * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int dir = (int)(i / ((long long)B * H));
        const long long rem = i % ((long long)B * H);
        const int b = (int)(rem / H), k = (int)(rem % H);
        const int slot_h0 = (dir == 0) ? 0 : T + 1, slot_other = (dir == 0) ? T + 1 : 0;
        const int ldy = ndir * H;
        y_ext[((long long)slot_h0 * B + b) * ldy + dir * H + k] = h0 ? h0[i] : 0.f;
        y_ext[((long long)slot_other * B + b) * ldy + dir * H + k] = 0.f;
    }
}

// Backward, one launch per step (reverse processing order).  GEMM part: M = dgh_{next} W_hh with
// dgh_{next} = [dgi_next(:, 0:2H) | dghn_next] (two k segments) on 64 x 128 tiles; epilogue: dh = dy_t +
// keep_next + M, gate gradients of step t -> dgi_t, dghn_t, keep_t (= z * dh).  first = 1: no GEMM, the
// running gradient starts from dhn (or zero).  final = 1: only dh0 = keep + M is written.
struct GruStepBwd {
    const float* dy;        // (T, B, ndir*H) or null
    const float* dhn;       // (ndir, B, H) or null
    const float* y_ext;
    const float* saved;
    const float* w_hh[2];   // (3H x H): B operand [k][n]
    float* dgi;             // (ndir, T, B, 3H)
    float* dghn;            // (ndir, T, B, H)
    float* keep_in;         // (ndir, B, H)  z*dh of the step processed before (next in time order)
    float* keep_out;
    float* dh0;             // (ndir, B, H), final pass only
    int T, B, H, ndir, s, first, final;
    int vecA1, vecA2, vecB;
};

template <bool BF>
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(GruStepBwd p) {
    using namespace xps_tile;
    __shared__ __attribute__((aligned(16))) TileMem<BF, 1> mem;
    const int dir = blockIdx.z, H = p.H, B = p.B, T = p.T;
    const int ldy = p.ndir * H;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[1][2];
    zero_acc<1>(acc);
    if (!p.first) {
        // step processed just before this one (one step LATER in processing order)
        const int sn = p.final ? 0 : p.s + 1;
        const int tn = (dir == 0) ? sn : T - 1 - sn;
        const float* A1 = p.dgi + (((long long)dir * T + tn) * B) * 3 * H;       // (B x 2H of 3H)
        const float* A2 = p.dghn + (((long long)dir * T + tn) * B) * H;          // (B x H)
        RowMap r1; r1.gs = 0; r1.ld = 3 * H; r1.rpg = 1 << 30;
        RowMap r2; r2.gs = 0; r2.ld = H; r2.rpg = 1 << 30;
        RowMap rb; rb.gs = 0; rb.ld = H; rb.rpg = 1 << 30;
        float nocs = 0.f;
        f32x4 nocs4 = {0.f, 0.f, 0.f, 0.f};
        const float* W = p.w_hh[dir];
        gemm_accumulate_any<true, false, 1, false, BF>(acc, nocs, nocs4, false, A1, r1, W, rb, B, H, 2 * H, m0, n0, 0, 2 * H, p.vecA1, p.vecB, mem);
        gemm_accumulate_any<true, false, 1, false, BF>(acc, nocs, nocs4, false, A2, r2, W + (long long)2 * H * H, rb, B, H, H, m0, n0, 0, H, p.vecA2, p.vecB, mem);
    }
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;
    const int li = lane & 31, lk = lane >> 5;
    const int t = (dir == 0) ? p.s : T - 1 - p.s;
    const int slot_prev = (dir == 0) ? t : t + 2;
    // inputs and outputs of the gate-gradient epilogue never overlap; without the promise the stores of one row order the
    // loads of the next and the 16 rows become 16 dependent global round trips.  All inputs of a 32-unit column block are
    // requested first (7 loads x 16 rows), then the rows are computed and stored.
    const float* __restrict__ in_dhn = p.dhn;
    const float* __restrict__ in_keep = p.keep_in;
    const float* __restrict__ in_dy = p.dy;
    const float* __restrict__ in_saved = p.saved;
    const float* __restrict__ in_y = p.y_ext;
    float* __restrict__ out_dgi = p.dgi;
    float* __restrict__ out_dghn = p.dghn;
    float* __restrict__ out_keep = p.keep_out;
    float* __restrict__ out_dh0 = p.dh0;
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
        const int j = n0 + wn + jn * 32 + li;
        if (j >= H) continue;
        float v_base[16], v_dy[16], v_r[16], v_z[16], v_n[16], v_q[16], v_hp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const bool ok = b < B;
            const int bc = ok ? b : 0;
            const long long ob = ((long long)dir * B + bc) * H + j;
            v_base[r] = p.first ? (in_dhn ? in_dhn[ob] : 0.f) : in_keep[ob];
            if (!p.final) {
                v_dy[r] = in_dy ? in_dy[((long long)t * B + bc) * ldy + dir * H + j] : 0.f;
                const float* sv = in_saved + (((long long)dir * T + t) * B + bc) * 4 * H;
                v_r[r] = sv[j]; v_z[r] = sv[H + j]; v_n[r] = sv[2 * H + j]; v_q[r] = sv[3 * H + j];
                v_hp[r] = in_y[((long long)slot_prev * B + bc) * ldy + dir * H + j];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (b >= B) continue;
            const long long ob = ((long long)dir * B + b) * H + j;
            float dh = p.first ? v_base[r] : acc[0][jn][r] + v_base[r];
            if (p.final) { out_dh0[ob] = dh; continue; }
            dh += v_dy[r];
            const float rg = v_r[r], zg = v_z[r], ng = v_n[r], q = v_q[r], hp = v_hp[r];
            const float dn = dh * (1.f - zg);
            const float dz = dh * (hp - ng);
            const float dan = dn * (1.f - ng * ng);
            const long long o = (((long long)dir * T + t) * B + b) * 3 * H;
            out_dgi[o + j] = dan * q * rg * (1.f - rg);
            out_dgi[o + H + j] = dz * zg * (1.f - zg);
            out_dgi[o + 2 * H + j] = dan;
            out_dghn[(((long long)dir * T + t) * B + b) * H + j] = dan * rg;
            out_keep[ob] = dh * zg;
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

// up to 4 equally shaped matrices in one launch (blockIdx.z selects the matrix)
struct TransposeBatch { const float* src[4]; float* dst[4]; };
__global__ void transpose_batched_kernel(TransposeBatch tb, int rows, int cols) {
    __shared__ float tile[32][33];
    const float* __restrict__ src = tb.src[blockIdx.z];
    float* __restrict__ dst = tb.dst[blockIdx.z];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int r = by + i, c = bx + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? src[(long long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        int c = bx + i, r = by + threadIdx.x;
        if (c < cols && r < rows) dst[(long long)c * rows + r] = tile[threadIdx.x][i];
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// XPS_GRU_STEP_PATH=0 forces the streaming persistent kernels for H > 128 (A/B comparisons)
inline bool use_step_path() {
    static const bool on = [] { const char* e = getenv("XPS_GRU_STEP_PATH"); return !(e && e[0] == '0'); }();
    return on;
}

}  // namespace

// The cluster kernels address y_ext, saved and dgi through raw buffer descriptors (32-bit sizes and offsets): the LARGEST of
// them, saved = ndir * T * B * 4H floats (y_ext is (T + 2) * B * ndir * H, dgi ndir * T * B * 3H), must stay below 4 GiB;
// longer sequences take the per-step path.
static bool cluster_shape_ok(int T, int B, int H, int ndir) {
    return xps_internal_gru_cluster_usable(B, H, ndir) && (long long)ndir * (T + 2) * B * 4 * H * 4 < (1ll << 32);
}

extern "C" size_t xps_gru_seq_fwd_f32_workspace(int T, int B, int H, int ndir) {
    if (T <= 0 || B <= 0 || H <= 0 || (ndir != 1 && ndir != 2)) return 16;
    return cluster_shape_ok(T, B, H, ndir) ? xps_internal_gru_cluster_fwd_workspace(B, H, ndir) : 16;
}

extern "C" long long xps_gru_seq_status_offset(int T, int B, int H, int ndir) {
    if (T <= 0 || B <= 0 || H <= 0 || (ndir != 1 && ndir != 2)) return -1;
    return cluster_shape_ok(T, B, H, ndir) ? (long long)xps_internal_gru_cluster_status_offset(B, H, ndir) : -1;
}

// the shapes whose forward / backward run the register-resident kernels (the only ones that fuse the inter-layer dropout)
extern "C" int xps_gru_seq_fused_dropout_supported(int T, int B, int H, int ndir) {
    if (T < 1 || B < 1 || (ndir != 1 && ndir != 2)) return 0;
    if (cluster_shape_ok(T, B, H, ndir)) return 0;
    return (H == 128 || H == 64) ? 1 : 0;
}

static int gru_seq_fwd_impl(const float* gi, const float* const* w_hh, const float* const* b_hh,
                            const float* h0, float* y_ext, float* saved,
                            int T, int B, int H, int ndir, float* y_drop, float drop_p, unsigned long long drop_seed,
                            void* workspace, size_t workspace_bytes, void* stream);

extern "C" int xps_gru_seq_fwd_f32(const float* gi, const float* const* w_hh, const float* const* b_hh,
                                   const float* h0, float* y_ext, float* saved,
                                   int T, int B, int H, int ndir, void* workspace, size_t workspace_bytes, void* stream) {
    return gru_seq_fwd_impl(gi, w_hh, b_hh, h0, y_ext, saved, T, B, H, ndir, nullptr, 0.f, 0ull, workspace, workspace_bytes, stream);
}

extern "C" int xps_gru_seq_fwd_drop_f32(const float* gi, const float* const* w_hh, const float* const* b_hh,
                                        const float* h0, float* y_ext, float* saved,
                                        int T, int B, int H, int ndir, float* y_drop, float drop_p, uint64_t drop_seed,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(y_drop && drop_p >= 0.f && drop_p < 1.f, "y_drop must be given, 0 <= p < 1");
    XPS_CHECK_ARG(xps_gru_seq_fused_dropout_supported(T, B, H, ndir) && aligned16(y_drop),
                  "fused dropout: shape not on the resident kernels (see xps_gru_seq_fused_dropout_supported)");
    return gru_seq_fwd_impl(gi, w_hh, b_hh, h0, y_ext, saved, T, B, H, ndir, y_drop, drop_p, (unsigned long long)drop_seed, workspace,
                            workspace_bytes, stream);
}

// the cluster-persistent shapes in bf16x3 mode: the forward kernel's epilogue can also write XPS_FMT_SPLIT4 images of its outputs
extern "C" int xps_gru_seq_fwd_images_supported(int T, int B, int H, int ndir) {
    if (T < 1 || B < 1 || H < 1 || (ndir != 1 && ndir != 2)) return 0;
    return (cluster_shape_ok(T, B, H, ndir) && xps_internal_gemm_mode() == 1 && H % 4 == 0) ? 1 : 0;
}

// ... and the image of y_ext costs nothing there: the launch uses it as its exchange buffer (H = 512, no pad trials; XPS_GRU_XIMG=0: never)
extern "C" int xps_gru_seq_fwd_image_exchange_supported(int T, int B, int H, int ndir) {
    return (xps_gru_seq_fwd_images_supported(T, B, H, ndir) && xps_internal_gru_cluster_ximg_ok(B, H, ndir)) ? 1 : 0;
}

extern "C" int xps_gru_seq_fwd_images_f32(const float* gi, const float* const* w_hh, const float* const* b_hh,
                                          const float* h0, float* y_ext, float* saved, int T, int B, int H, int ndir,
                                          float* y_split, float* y_drop_split, float drop_p, uint64_t drop_seed,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(gi && w_hh && b_hh && y_ext, "null argument");
    XPS_CHECK_ARG(T >= 1 && B >= 1 && H >= 1 && (ndir == 1 || ndir == 2), "bad shape");
    XPS_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "0 <= p < 1");
    XPS_CHECK_ARG(xps_gru_seq_fwd_images_supported(T, B, H, ndir),
                  "XPS_FMT_SPLIT4 images: cluster-persistent shapes in bf16x3 mode only (see xps_gru_seq_fwd_images_supported)");
    XPS_CHECK_ARG(w_hh[0] && b_hh[0] && (ndir == 1 || (w_hh[1] && b_hh[1])), "null weight pointer");
    if (!workspace || workspace_bytes < xps_gru_seq_fwd_f32_workspace(T, B, H, ndir)) {
        xps_set_error("xps_gru_seq_fwd_images_f32: workspace too small");
        return XPS_E_WORKSPACE;
    }
    return xps_internal_gru_cluster_fwd(gi, w_hh, b_hh, h0, y_ext, saved, T, B, H, ndir, workspace, (hipStream_t)stream,
                                        y_split, y_drop_split, drop_p, (unsigned long long)drop_seed);
}

static int gru_seq_fwd_impl(const float* gi, const float* const* w_hh, const float* const* b_hh,
                            const float* h0, float* y_ext, float* saved,
                            int T, int B, int H, int ndir, float* y_drop, float drop_p, unsigned long long drop_seed,
                            void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(gi && w_hh && b_hh && y_ext, "null argument");
    XPS_CHECK_ARG(T >= 1 && B >= 1 && H >= 1, "T, B, H must be >= 1");
    XPS_CHECK_ARG(ndir == 1 || ndir == 2, "ndir must be 1 or 2");
    if (cluster_shape_ok(T, B, H, ndir)) {
        XPS_CHECK_ARG(w_hh[0] && b_hh[0] && (ndir == 1 || (w_hh[1] && b_hh[1])), "null weight pointer");
        if (!workspace || workspace_bytes < xps_gru_seq_fwd_f32_workspace(T, B, H, ndir)) {
            xps_set_error("xps_gru_seq_fwd_f32: workspace too small");
            return XPS_E_WORKSPACE;
        }
        return xps_internal_gru_cluster_fwd(gi, w_hh, b_hh, h0, y_ext, saved, T, B, H, ndir, workspace, (hipStream_t)stream,
                                            nullptr, nullptr, 0.f, 0ull);
    }
    GruFwdParams p;
    p.gi = gi; p.h0 = h0; p.y_ext = y_ext; p.saved = saved;
    p.T = T; p.B = B; p.H = H; p.ndir = ndir;
    p.y_drop = y_drop; p.drop_p = drop_p; p.drop_scale = 1.0f / (1.0f - drop_p); p.drop_seed = drop_seed;
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
    XPS_CHECK_ARG(!y_drop || vec, "fused dropout needs 16-byte aligned weights");
    if (H > 128 && use_step_path()) {
        // large hidden size: one fused GEMM + gate launch per time step (see gru_step_fwd_kernel)
        hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL(gru_init_slots_kernel, dim3(cdiv((long long)ndir * B * H, 256) > 1024 ? 1024 : cdiv((long long)ndir * B * H, 256)),
                           dim3(256), 0, st, h0, y_ext, T, B, H, ndir);
        GruStepFwd q;
        q.gi = gi; q.y_ext = y_ext; q.saved = saved; q.T = T; q.B = B; q.H = H; q.ndir = ndir;
        for (int d = 0; d < 2; ++d) { q.w_hh[d] = p.w_hh[d]; q.b_hh[d] = p.b_hh[d]; }
        q.vecA = (int)(H % 4 == 0 && aligned16(y_ext));
        q.vecB = (int)vec;
        dim3 sgrid(cdiv(H, 32), cdiv(B, 128), ndir);
        for (int s = 0; s < T; ++s) {
            q.s = s;
            if (xps_internal_gemm_mode() == 1) hipLaunchKernelGGL(gru_step_fwd_kernel<true>, sgrid, dim3(256), 0, st, q);
            else hipLaunchKernelGGL(gru_step_fwd_kernel<false>, sgrid, dim3(256), 0, st, q);
        }
        XPS_CHECK_LAUNCH();
        return XPS_OK;
    }
    dim3 grid(cdiv(B, GBM), ndir);
    if (vec && (H == 128 || H == 64)) {
        const bool bf = xps_internal_gemm_mode() == 1;
        if (H == 128 && bf) hipLaunchKernelGGL((gru_fwd_resident_kernel<128, true, GRU_BF_WAVES>), grid, dim3(GRU_BF_WAVES * 64), 0, (hipStream_t)stream, p);
        else if (H == 128) hipLaunchKernelGGL((gru_fwd_resident_kernel<128, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else if (bf) hipLaunchKernelGGL((gru_fwd_resident_kernel<64, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((gru_fwd_resident_kernel<64, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
        XPS_CHECK_LAUNCH();
        return XPS_OK;
    }
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

extern "C" size_t xps_gru_seq_bwd_f32_workspace(int T, int B, int H, int ndir) {
    (void)T;
    if (B <= 0 || H <= 0 || ndir <= 0) return 16;
    if (T > 0 && (ndir == 1 || ndir == 2) && cluster_shape_ok(T, B, H, ndir)) return xps_internal_gru_cluster_bwd_workspace(B, H, ndir);
    return (size_t)2 * ndir * B * H * sizeof(float) + 16;       // ping-pong running gradient of the per-step path
}

static int gru_seq_bwd_impl(const float* dy, const float* dhn, const float* y_ext, const float* saved,
                            const float* const* w_hh, const float* const* w_hh_t, float* dgi, float* dghn, float* dh0,
                            int T, int B, int H, int ndir, int has_drop, float drop_p, unsigned long long drop_seed,
                            void* workspace, size_t workspace_bytes, void* stream, int split_out = 0);

extern "C" int xps_gru_seq_bwd_f32(const float* dy, const float* dhn, const float* y_ext, const float* saved,
                                   const float* const* w_hh, const float* const* w_hh_t, float* dgi, float* dghn, float* dh0,
                                   int T, int B, int H, int ndir, void* workspace, size_t workspace_bytes, void* stream) {
    return gru_seq_bwd_impl(dy, dhn, y_ext, saved, w_hh, w_hh_t, dgi, dghn, dh0, T, B, H, ndir, 0, 0.f, 0ull, workspace, workspace_bytes, stream);
}

extern "C" int xps_gru_seq_bwd_drop_f32(const float* dy, const float* dhn, const float* y_ext, const float* saved,
                                        const float* const* w_hh, const float* const* w_hh_t, float* dgi, float* dghn, float* dh0,
                                        int T, int B, int H, int ndir, float drop_p, uint64_t drop_seed,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(dy && drop_p >= 0.f && drop_p < 1.f, "dy must be given, 0 <= p < 1");
    XPS_CHECK_ARG(xps_gru_seq_fused_dropout_supported(T, B, H, ndir),
                  "fused dropout: shape not on the resident kernels (see xps_gru_seq_fused_dropout_supported)");
    return gru_seq_bwd_impl(dy, dhn, y_ext, saved, w_hh, w_hh_t, dgi, dghn, dh0, T, B, H, ndir, 1, drop_p, (unsigned long long)drop_seed,
                            workspace, workspace_bytes, stream);
}

extern "C" int xps_gru_seq_bwd_split4_supported(int T, int B, int H, int ndir) {
    if (T < 1 || B < 1 || (ndir != 1 && ndir != 2) || xps_internal_gemm_mode() != 1) return 0;
    if (cluster_shape_ok(T, B, H, ndir)) return 1;
    return xps_gru_seq_fused_dropout_supported(T, B, H, ndir);        // the register-resident kernels (H = 64 / 128)
}

extern "C" int xps_gru_seq_bwd_split4_f32(const float* dy, const float* dhn, const float* y_ext, const float* saved,
                                          const float* const* w_hh, const float* const* w_hh_t, float* dgi, float* dghn, float* dh0,
                                          int T, int B, int H, int ndir, float drop_p, uint64_t drop_seed,
                                          void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "0 <= p < 1");
    XPS_CHECK_ARG(drop_p == 0.f || (dy && xps_gru_seq_fused_dropout_supported(T, B, H, ndir) && !cluster_shape_ok(T, B, H, ndir)),
                  "fused dropout: dy must be given and the shape must be on the resident kernels");
    return gru_seq_bwd_impl(dy, dhn, y_ext, saved, w_hh, w_hh_t, dgi, dghn, dh0, T, B, H, ndir, drop_p > 0.f ? 1 : 0, drop_p,
                            (unsigned long long)drop_seed, workspace, workspace_bytes, stream, 1);
}

static int gru_seq_bwd_impl(const float* dy, const float* dhn, const float* y_ext, const float* saved,
                            const float* const* w_hh, const float* const* w_hh_t, float* dgi, float* dghn, float* dh0,
                            int T, int B, int H, int ndir, int has_drop, float drop_p, unsigned long long drop_seed,
                            void* workspace, size_t workspace_bytes, void* stream, int split_out) {
    XPS_CHECK_ARG(y_ext && saved && w_hh && w_hh_t && dgi && dghn, "null argument");
    XPS_CHECK_ARG(!split_out || xps_gru_seq_bwd_split4_supported(T, B, H, ndir),
                  "XPS_FMT_SPLIT4 outputs: shape / precision mode not served (see xps_gru_seq_bwd_split4_supported)");
    XPS_CHECK_ARG(dy || dhn, "at least one of dy / dhn must be given");
    XPS_CHECK_ARG(T >= 1 && B >= 1 && H >= 1, "T, B, H must be >= 1");
    XPS_CHECK_ARG(ndir == 1 || ndir == 2, "ndir must be 1 or 2");
    if (cluster_shape_ok(T, B, H, ndir)) {
        XPS_CHECK_ARG(w_hh_t[0] && (ndir == 1 || w_hh_t[1]), "null weight pointer");
        if (!workspace || workspace_bytes < xps_gru_seq_bwd_f32_workspace(T, B, H, ndir)) {
            xps_set_error("xps_gru_seq_bwd_f32: workspace too small");
            return XPS_E_WORKSPACE;
        }
        return xps_internal_gru_cluster_bwd(dy, dhn, y_ext, saved, w_hh_t, dgi, dghn, dh0, T, B, H, ndir, workspace, (hipStream_t)stream, split_out);
    }
    GruBwdParams p;
    p.dy = dy; p.dhn = dhn; p.y_ext = y_ext; p.saved = saved; p.dgi = dgi; p.dghn = dghn; p.dh0 = dh0;
    p.T = T; p.B = B; p.H = H; p.ndir = ndir;
    p.has_drop = has_drop; p.drop_p = drop_p; p.drop_scale = 1.0f / (1.0f - drop_p); p.drop_seed = drop_seed;
    p.split_out = split_out;
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
    XPS_CHECK_ARG(!has_drop || vec, "fused dropout needs 16-byte aligned weights");
    XPS_CHECK_ARG(!split_out || (vec && (H == 128 || H == 64)), "XPS_FMT_SPLIT4 outputs need the resident kernels (16-byte aligned weights)");
    if (H > 128 && use_step_path()) {
        if (!workspace || workspace_bytes < xps_gru_seq_bwd_f32_workspace(T, B, H, ndir) || !aligned16(workspace)) {
            xps_set_error("xps_gru_seq_bwd_f32: workspace too small or misaligned");
            return XPS_E_WORKSPACE;
        }
        hipStream_t st = (hipStream_t)stream;
        GruStepBwd q;
        q.dy = dy; q.dhn = dhn; q.y_ext = y_ext; q.saved = saved; q.dgi = dgi; q.dghn = dghn; q.dh0 = dh0;
        q.T = T; q.B = B; q.H = H; q.ndir = ndir;
        bool vb = (H % 4 == 0);
        for (int d = 0; d < 2; ++d) {
            q.w_hh[d] = w_hh[d < ndir ? d : 0];
            XPS_CHECK_ARG(q.w_hh[d], "null weight pointer");
            vb = vb && aligned16(q.w_hh[d]);
        }
        q.vecB = (int)vb;
        q.vecA1 = (int)(H % 4 == 0 && aligned16(dgi));
        q.vecA2 = (int)(H % 4 == 0 && aligned16(dghn));
        float* keep[2] = {(float*)workspace, (float*)workspace + (size_t)ndir * B * H};
        dim3 sgrid(cdiv(H, 128), cdiv(B, 64), ndir);
        int pp = 0;
        for (int s = T - 1; s >= 0; --s) {
            q.s = s; q.first = (s == T - 1); q.final = 0;
            q.keep_in = keep[pp]; q.keep_out = keep[pp ^ 1];
            if (xps_internal_gemm_mode() == 1) hipLaunchKernelGGL(gru_step_bwd_kernel<true>, sgrid, dim3(256), 0, st, q);
            else hipLaunchKernelGGL(gru_step_bwd_kernel<false>, sgrid, dim3(256), 0, st, q);
            pp ^= 1;
        }
        if (dh0) {
            q.s = 0; q.first = 0; q.final = 1; q.keep_in = keep[pp]; q.keep_out = keep[pp ^ 1];
            if (xps_internal_gemm_mode() == 1) hipLaunchKernelGGL(gru_step_bwd_kernel<true>, sgrid, dim3(256), 0, st, q);
            else hipLaunchKernelGGL(gru_step_bwd_kernel<false>, sgrid, dim3(256), 0, st, q);
        }
        XPS_CHECK_LAUNCH();
        return XPS_OK;
    }
    dim3 grid(cdiv(B, GBM), ndir);
    if (vec && (H == 128 || H == 64)) {
        const bool bf = xps_internal_gemm_mode() == 1;
        if (H == 128 && bf) hipLaunchKernelGGL((gru_bwd_resident_kernel<128, true, GRU_BF_WAVES>), grid, dim3(GRU_BF_WAVES * 64), 0, (hipStream_t)stream, p);
        else if (H == 128) hipLaunchKernelGGL((gru_bwd_resident_kernel<128, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else if (bf) hipLaunchKernelGGL((gru_bwd_resident_kernel<64, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((gru_bwd_resident_kernel<64, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
        XPS_CHECK_LAUNCH();
        return XPS_OK;
    }
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

#ifdef XPS_STAMP
extern "C" int xps_debug_read_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int xps_transpose_f32(const float* src, float* dst, int rows, int cols, void* stream) {
    XPS_CHECK_ARG(src && dst && rows >= 0 && cols >= 0, "bad argument");
    if (rows == 0 || cols == 0) return XPS_OK;
    dim3 grid(cdiv(cols, 32), cdiv(rows, 32));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, src, dst, rows, cols);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_transpose_batched_f32(const float* const* src, float* const* dst, int n, int rows, int cols, void* stream) {
    XPS_CHECK_ARG(src && dst && n >= 1 && n <= 4 && rows >= 0 && cols >= 0, "bad argument (1..4 matrices)");
    if (rows == 0 || cols == 0) return XPS_OK;
    TransposeBatch tb;
    for (int i = 0; i < 4; ++i) {
        const int j = i < n ? i : 0;
        XPS_CHECK_ARG(src[j] && dst[j], "null matrix pointer");
        tb.src[i] = src[j]; tb.dst[i] = dst[j];
    }
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32), n), dim3(32, 8), 0, (hipStream_t)stream,
                       tb, rows, cols);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}
