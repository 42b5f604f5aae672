// Fused autoregressive decoder (reference nn_models/models.py:285-299 loop around DecoderRNN
// :758-761): for every decode step  embedding/input-projection gather -> GRU cell (W_hh resident
// in VGPRs, h W_hh^T on the f32 MFMA or, in the default precision, as bf16 split products) -> Linear(H, n_classes) ->
// argmax-or-teacher next token,
// ALL steps in ONE persistent launch: no host round trip per step (the reference syncs on
// `torch.rand(1).item()` and `argmax` every step) and no per-step kernel boundaries.
//
// One workgroup = 16 trials; lane (n = l & 15, kq = l >> 4) of wave w owns hidden units
// (w + 4*tt)*16 + 4*kq + {0..3} of trial n (same operand-swapped MFMA mapping as the resident
// encoder kernels in xps_gru.hip).  One-layer decoders with H = 64 or 128 only; other shapes go
// through the composed path (gather + xps_gru_seq + GEMM).
#include "xps_common.h"
#include "xps_gemm_tile.h"
using xps_tile::bf16x4;
using xps_tile::bf16x8;
using xps_tile::bf_split;

namespace {

constexpr int DBM = 16;      // trials per workgroup
constexpr int MAXC = 16;     // max classes
constexpr int MAXL = 8;      // max decode steps

__device__ inline float d_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ inline float d_tanh(float x) { return 2.0f * d_sigmoid(2.0f * x) - 1.0f; }
__device__ inline float g4(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

struct DecFwdParams {
    const float* table;     // (ntok x 3H)  E W_ih^T + b_ih
    const float* w_hh;      // (3H x H)
    const float* b_hh;      // (3H)
    const float* h0;        // (B x H)
    const float* w_fc;      // (C x H)
    const float* b_fc;      // (C)
    const long long* teacher;   // (B x L) or null
    const int* flags;       // (L) device flags: 1 = feed the teacher token after this step
    float* logits;          // (B x L x C)
    long long* tokens;      // (L x B) input token of every step
    float* hs;              // ((L+1) x B x H)
    float* saved;           // (L x B x 4H) or null
    int B, C, L, ntok, start_token;
};

// BF: the recurrent product on the bf16 matrix pipe with split operands, as in the GRU sequence kernels (xps_gru.hip)
template <int H, bool BF>
__global__ __launch_bounds__(256, 1) void decoder_fwd_kernel(DecFwdParams p) {
    constexpr int NT = H / 16, TPW = NT / 4, NC = H / 16, LDH = H + 4;
    constexpr int NCB = H / 32, LDB = H + 8;
    __shared__ __attribute__((aligned(16))) float hs_l[2][DBM][LDH];
    __shared__ __attribute__((aligned(16))) __bf16 hsb[BF ? 2 : 1][2][BF ? DBM : 1][BF ? LDB : 8];   // [buffer][hi, lo][trial][k]
    __shared__ float wfc[MAXC][H + 1];          // +1: the FC dot products read a column of classes
    __shared__ float lg[DBM][MAXC];
    __shared__ int tok_l[DBM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int b0 = blockIdx.x * DBM;
    const int B = p.B, C = p.C, L = p.L;
    const int b = b0 + n;
    const bool live = b < B;
    const int bc = live ? b : B - 1;

    float w[BF ? 1 : TPW][3][NC][4];
    bf16x8 wh[BF ? TPW : 1][3][NCB], wl[BF ? TPW : 1][3][NCB];
    float4 bias[TPW][3];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int j0 = (wave + 4 * tt) * 16;
            bias[tt][g] = *reinterpret_cast<const float4*>(p.b_hh + g * H + j0 + 4 * kq);
            if constexpr (BF) {
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    const float* wp = p.w_hh + (long long)(g * H + j0 + n) * H + 32 * c + 8 * kq;
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
                    const float4 v = *reinterpret_cast<const float4*>(p.w_hh + (long long)(g * H + j0 + n) * H + 16 * c + 4 * kq);
                    w[tt][g][c][0] = v.x; w[tt][g][c][1] = v.y; w[tt][g][c][2] = v.z; w[tt][g][c][3] = v.w;
                }
            }
        }
    for (int i = tid; i < C * H; i += 256) wfc[i / H][i % H] = p.w_fc[i];
    for (int i = tid; i < 2 * DBM * LDH; i += 256) (&hs_l[0][0][0])[i] = 0.f;
    if (tid < DBM) tok_l[tid] = p.start_token;
    __syncthreads();
    for (int i = tid; i < DBM * H; i += 256) {
        const int r = i / H, k = i % H, bb = b0 + r;
        if (bb < B) {
            const float v = p.h0[(long long)bb * H + k];
            hs_l[0][r][k] = v;
            p.hs[(long long)bb * H + k] = v;
        }
    }
    __syncthreads();
    if constexpr (BF) {
        for (int i = tid; i < DBM * H; i += 256) {
            const int r = i / H, k = i % H;
            __bf16 a, b;
            bf_split(hs_l[0][r][k], a, b);
            hsb[0][0][r][k] = a; hsb[0][1][r][k] = b;
            hsb[1][0][r][k] = (__bf16)0.f; hsb[1][1][r][k] = (__bf16)0.f;
        }
        __syncthreads();
    }

    for (int s = 0; s < L; ++s) {
        const int cur = s & 1;
        int tok = tok_l[n];
        tok = tok < 0 ? 0 : (tok >= p.ntok ? p.ntok - 1 : tok);
        if (live && wave == 0 && kq == 0) p.tokens[(long long)s * B + b] = tok;
        const float* gp = p.table + (long long)tok * 3 * H;
        float4 g_in[TPW][3];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                g_in[tt][g] = *reinterpret_cast<const float4*>(gp + g * H + (wave + 4 * tt) * 16 + 4 * kq);
        f32x4 acc[TPW][3];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[tt][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (BF) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const bf16x8 bh8 = *reinterpret_cast<const bf16x8*>(&hsb[cur][0][n][32 * c + 8 * kq]);
                const bf16x8 bl8 = *reinterpret_cast<const bf16x8*>(&hsb[cur][1][n][32 * c + 8 * kq]);
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        acc[tt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[tt][g][c], bh8, acc[tt][g], 0, 0, 0);
                        acc[tt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[tt][g][c], bl8, acc[tt][g], 0, 0, 0);
                        acc[tt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[tt][g][c], bh8, acc[tt][g], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 a4 = *reinterpret_cast<const float4*>(&hs_l[cur][n][16 * c + 4 * kq]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc[tt][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tt][g][c][e], g4(a4, e), acc[tt][g], 0, 0, 0);
        }
        }
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int j = (wave + 4 * tt) * 16 + 4 * kq;
            const float4 hp = *reinterpret_cast<const float4*>(&hs_l[cur][n][j]);
            float o[4], r_[4], z_[4], n_[4], q_[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float rg = d_sigmoid(g4(g_in[tt][0], i) + acc[tt][0][i] + g4(bias[tt][0], i));
                const float zg = d_sigmoid(g4(g_in[tt][1], i) + acc[tt][1][i] + g4(bias[tt][1], i));
                const float q = acc[tt][2][i] + g4(bias[tt][2], i);
                const float ng = d_tanh(g4(g_in[tt][2], i) + rg * q);
                o[i] = live ? ng + zg * (g4(hp, i) - ng) : 0.f;
                r_[i] = rg; z_[i] = zg; n_[i] = ng; q_[i] = q;
            }
            const float4 h4 = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(&hs_l[cur ^ 1][n][j]) = h4;
            if constexpr (BF) {
                bf16x4 sh, sl;
#pragma unroll
                for (int i = 0; i < 4; ++i) { __bf16 a, b; bf_split(o[i], a, b); sh[i] = a; sl[i] = b; }
                *reinterpret_cast<bf16x4*>(&hsb[cur ^ 1][0][n][j]) = sh;
                *reinterpret_cast<bf16x4*>(&hsb[cur ^ 1][1][n][j]) = sl;
            }
            if (live) {
                *reinterpret_cast<float4*>(p.hs + ((long long)(s + 1) * B + b) * H + j) = h4;
                if (p.saved) {
                    float* sv = p.saved + ((long long)s * B + b) * 4 * H;
                    *reinterpret_cast<float4*>(sv + j) = make_float4(r_[0], r_[1], r_[2], r_[3]);
                    *reinterpret_cast<float4*>(sv + H + j) = make_float4(z_[0], z_[1], z_[2], z_[3]);
                    *reinterpret_cast<float4*>(sv + 2 * H + j) = make_float4(n_[0], n_[1], n_[2], n_[3]);
                    *reinterpret_cast<float4*>(sv + 3 * H + j) = make_float4(q_[0], q_[1], q_[2], q_[3]);
                }
            }
        }
        __syncthreads();
        // logits = h_new W_fc^T + b_fc : thread -> (trial r, class c)
        if (tid < DBM * C) {
            const int r = tid / C, c = tid % C;
            float a = p.b_fc[c];
            const float* hrow = &hs_l[cur ^ 1][r][0];
#pragma unroll 8
            for (int k = 0; k < H; ++k) a = fmaf(hrow[k], wfc[c][k], a);
            lg[r][c] = a;
            if (b0 + r < B) p.logits[((long long)(b0 + r) * L + s) * C + c] = a;
        }
        __syncthreads();
        // next input token: teacher (device flag) or first-max argmax, as torch.argmax
        if (tid < DBM && s + 1 < L) {
            const int bb = b0 + tid;
            int t = 0;
            if (p.teacher && p.flags && p.flags[s]) {
                t = (int)p.teacher[(long long)(bb < B ? bb : B - 1) * L + s];
            } else {
                float best = lg[tid][0];
                for (int c = 1; c < C; ++c)
                    if (lg[tid][c] > best) { best = lg[tid][c]; t = c; }
            }
            tok_l[tid] = t;
        }
        __syncthreads();
    }
}

struct DecBwdParams {
    const float* dlogits;   // (B x L x C)
    const float* hs;        // ((L+1) x B x H)
    const float* saved;     // (L x B x 4H)
    const float* w_hh_t;    // (H x 3H)
    const float* w_fc;      // (C x H)
    float* dgi;             // (L x B x 3H)
    float* dghn;            // (L x B x H)
    float* dh0;             // (B x H)
    int B, C, L;
};

template <int H, bool BF>
__global__ __launch_bounds__(256, 1) void decoder_bwd_kernel(DecBwdParams p) {
    constexpr int NT = H / 16, TPW = NT / 4, NC = 3 * H / 16, LDG = 3 * H + 4, LDC = H + 4;
    constexpr int NCB = 3 * H / 32, LDGB = 3 * H + 8;
    constexpr int H4 = H / 4, GPT = DBM * H4 / 256;
    __shared__ __attribute__((aligned(16))) float G[BF ? 1 : DBM][BF ? 4 : LDG];
    __shared__ __attribute__((aligned(16))) __bf16 Gb[2][BF ? DBM : 1][BF ? LDGB : 8];      // BF: [hi, lo][trial][k]
    __shared__ __attribute__((aligned(16))) float Cy[DBM][LDC];
    __shared__ __attribute__((aligned(16))) float wfc[MAXC][H];
    __shared__ float dl[DBM][MAXC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int b0 = blockIdx.x * DBM;
    const int B = p.B, C = p.C, L = p.L;

    float w[BF ? 1 : TPW][NC][4];
    bf16x8 wh[BF ? TPW : 1][NCB], wl[BF ? TPW : 1][NCB];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int j = (wave + 4 * tt) * 16 + n;
        if constexpr (BF) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const float* wp = p.w_hh_t + (long long)j * 3 * H + 32 * c + 8 * kq;
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
                const float4 v = *reinterpret_cast<const float4*>(p.w_hh_t + (long long)j * 3 * H + 16 * c + 4 * kq);
                w[tt][c][0] = v.x; w[tt][c][1] = v.y; w[tt][c][2] = v.z; w[tt][c][3] = v.w;
            }
        }
    }
    for (int i = tid; i < C * H; i += 256) wfc[i / H][i % H] = p.w_fc[i];
    if constexpr (BF) {
        for (int i = tid; i < 2 * DBM * LDGB; i += 256) (&Gb[0][0][0])[i] = (__bf16)0.f;
    } else {
        for (int i = tid; i < DBM * LDG; i += 256) (&G[0][0])[i] = 0.f;
    }
    for (int i = tid; i < DBM * LDC; i += 256) (&Cy[0][0])[i] = 0.f;
    __syncthreads();

    for (int s = L - 1; s >= 0; --s) {
        if (tid < DBM * C) {
            const int r = tid / C, c = tid % C;
            dl[r][c] = (b0 + r < B) ? p.dlogits[((long long)(b0 + r) * L + s) * C + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < GPT; ++e) {
            const int idx = tid + 256 * e;
            const int r = idx / H4, j = (idx % H4) * 4, b = b0 + r;
            float4 dar = make_float4(0.f, 0.f, 0.f, 0.f), daz = dar, danr = dar, keep = dar;
            if (b < B) {
                const float4 cy4 = *reinterpret_cast<const float4*>(&Cy[r][j]);
                float dh[4] = {cy4.x, cy4.y, cy4.z, cy4.w};
                for (int c = 0; c < C; ++c) {
                    const float d = dl[r][c];
                    const float4 wv = *reinterpret_cast<const float4*>(&wfc[c][j]);
                    dh[0] = fmaf(d, wv.x, dh[0]); dh[1] = fmaf(d, wv.y, dh[1]);
                    dh[2] = fmaf(d, wv.z, dh[2]); dh[3] = fmaf(d, wv.w, dh[3]);
                }
                const float* sv = p.saved + ((long long)s * B + b) * 4 * H;
                const float4 rg = *reinterpret_cast<const float4*>(sv + j);
                const float4 zg = *reinterpret_cast<const float4*>(sv + H + j);
                const float4 ng = *reinterpret_cast<const float4*>(sv + 2 * H + j);
                const float4 q = *reinterpret_cast<const float4*>(sv + 3 * H + j);
                const float4 hp = *reinterpret_cast<const float4*>(p.hs + ((long long)s * B + b) * H + j);
                float o_dar[4], o_daz[4], o_dan[4], o_danr[4], o_keep[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float r_ = g4(rg, i), z_ = g4(zg, i), n_ = g4(ng, i);
                    const float dn = dh[i] * (1.f - z_);
                    const float dz = dh[i] * (g4(hp, i) - n_);
                    const float dan = dn * (1.f - n_ * n_);
                    o_daz[i] = dz * z_ * (1.f - z_);
                    o_dar[i] = dan * g4(q, i) * r_ * (1.f - r_);
                    o_dan[i] = dan;
                    o_danr[i] = dan * r_;
                    o_keep[i] = dh[i] * z_;
                }
                dar = make_float4(o_dar[0], o_dar[1], o_dar[2], o_dar[3]);
                daz = make_float4(o_daz[0], o_daz[1], o_daz[2], o_daz[3]);
                danr = make_float4(o_danr[0], o_danr[1], o_danr[2], o_danr[3]);
                keep = make_float4(o_keep[0], o_keep[1], o_keep[2], o_keep[3]);
                const long long o = ((long long)s * B + b) * 3 * H;
                *reinterpret_cast<float4*>(p.dgi + o + j) = dar;
                *reinterpret_cast<float4*>(p.dgi + o + H + j) = daz;
                *reinterpret_cast<float4*>(p.dgi + o + 2 * H + j) = make_float4(o_dan[0], o_dan[1], o_dan[2], o_dan[3]);
                *reinterpret_cast<float4*>(p.dghn + ((long long)s * B + b) * H + j) = danr;
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
        __syncthreads();
        f32x4 acc[TPW];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const float4 c4 = *reinterpret_cast<const float4*>(&Cy[n][(wave + 4 * tt) * 16 + 4 * kq]);
            acc[tt] = (f32x4){c4.x, c4.y, c4.z, c4.w};
        }
        if constexpr (BF) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const bf16x8 bh8 = *reinterpret_cast<const bf16x8*>(&Gb[0][n][32 * c + 8 * kq]);
                const bf16x8 bl8 = *reinterpret_cast<const bf16x8*>(&Gb[1][n][32 * c + 8 * kq]);
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) {
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[tt][c], bh8, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[tt][c], bl8, acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[tt][c], bh8, acc[tt], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float4 a4 = *reinterpret_cast<const float4*>(&G[n][16 * c + 4 * kq]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt)
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tt][c][e], g4(a4, e), acc[tt], 0, 0, 0);
        }
        }
        const bool live = b0 + n < B;
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const float4 o = live ? make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&Cy[n][(wave + 4 * tt) * 16 + 4 * kq]) = o;
        }
        __syncthreads();
    }
    for (int i = tid; i < DBM * H; i += 256) {
        const int r = i / H, k = i % H, b = b0 + r;
        if (b < B) p.dh0[(long long)b * H + k] = Cy[r][k];
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int xps_decoder_supported(int H, int C, int L) {
    return (H == 64 || H == 128) && C >= 1 && C <= MAXC && L >= 1 && L <= MAXL;
}

extern "C" int xps_decoder_fwd_f32(const float* table, const float* w_hh, const float* b_hh, const float* h0,
                                   const float* w_fc, const float* b_fc, const int64_t* teacher, const int32_t* flags,
                                   float* logits, int64_t* tokens, float* hs, float* saved,
                                   int B, int H, int C, int L, int ntok, int start_token, void* stream) {
    XPS_CHECK_ARG(table && w_hh && b_hh && h0 && w_fc && b_fc && logits && tokens && hs, "null argument");
    XPS_CHECK_ARG(B >= 1 && ntok >= 1 && start_token >= 0 && start_token < ntok, "bad sizes");
    XPS_CHECK_ARG(xps_decoder_supported(H, C, L), "unsupported decoder shape (H in {64,128}, C <= 16, L <= 8)");
    XPS_CHECK_ARG(aligned16(table) && aligned16(w_hh) && aligned16(b_hh) && aligned16(hs) && (!saved || aligned16(saved)),
                  "pointers must be 16-byte aligned");
    DecFwdParams p;
    p.table = table; p.w_hh = w_hh; p.b_hh = b_hh; p.h0 = h0; p.w_fc = w_fc; p.b_fc = b_fc;
    p.teacher = (const long long*)teacher; p.flags = (const int*)flags;
    p.logits = logits; p.tokens = (long long*)tokens; p.hs = hs; p.saved = saved;
    p.B = B; p.C = C; p.L = L; p.ntok = ntok; p.start_token = start_token;
    dim3 grid(cdiv(B, DBM));
    const bool bf = xps_internal_gemm_mode() == 1;
    if (H == 128 && bf) hipLaunchKernelGGL((decoder_fwd_kernel<128, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (H == 128) hipLaunchKernelGGL((decoder_fwd_kernel<128, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (bf) hipLaunchKernelGGL((decoder_fwd_kernel<64, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((decoder_fwd_kernel<64, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_decoder_bwd_f32(const float* dlogits, const float* hs, const float* saved, const float* w_hh_t,
                                   const float* w_fc, float* dgi, float* dghn, float* dh0,
                                   int B, int H, int C, int L, void* stream) {
    XPS_CHECK_ARG(dlogits && hs && saved && w_hh_t && w_fc && dgi && dghn && dh0, "null argument");
    XPS_CHECK_ARG(B >= 1, "bad sizes");
    XPS_CHECK_ARG(xps_decoder_supported(H, C, L), "unsupported decoder shape (H in {64,128}, C <= 16, L <= 8)");
    XPS_CHECK_ARG(aligned16(hs) && aligned16(saved) && aligned16(w_hh_t) && aligned16(dgi) && aligned16(dghn),
                  "pointers must be 16-byte aligned");
    DecBwdParams p;
    p.dlogits = dlogits; p.hs = hs; p.saved = saved; p.w_hh_t = w_hh_t; p.w_fc = w_fc;
    p.dgi = dgi; p.dghn = dghn; p.dh0 = dh0; p.B = B; p.C = C; p.L = L;
    dim3 grid(cdiv(B, DBM));
    const bool bf = xps_internal_gemm_mode() == 1;
    if (H == 128 && bf) hipLaunchKernelGGL((decoder_bwd_kernel<128, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (H == 128) hipLaunchKernelGGL((decoder_bwd_kernel<128, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (bf) hipLaunchKernelGGL((decoder_bwd_kernel<64, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((decoder_bwd_kernel<64, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}
