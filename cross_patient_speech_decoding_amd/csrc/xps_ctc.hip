// CTC loss of the realtime CTC-RNN family (realtime_sim/realtime_nn_model.py:150 nn.CTCLoss(blank, zero_infinity=True),
// :213-224 log_softmax(2) -> criterion on (T, B, C)): log-softmax, the alpha / beta recursions and the gradient with
// respect to the LOGITS in one launch, one 64-lane workgroup per sample.
//
// The recursions are sequential in time and only 2L+1 wide (L = 3 phonemes -> 7 states), so the work per
// sample is a few hundred dependent steps: the kernel is latency-bound by construction; what matters is that a
// whole batch is one launch (the samples run in parallel across the chip) and that no (T, B, C) log-prob tensor is
// materialised: lp[t][c] = logit - lse[t] is recomputed from the logits.
//
// Formulas follow torch.nn.functional.ctc_loss (LossCTC.cpp): extended label l' (blank, l1, blank, ..., blank),
// alpha_t(s) = lp_t(l'_s) + logsumexp(alpha_{t-1}(s), alpha_{t-1}(s-1), [l'_s != l'_{s-2}] alpha_{t-1}(s-2)),
// nll = -logsumexp(alpha_{T-1}(S-1), alpha_{T-1}(S-2)); d nll / d logit_t(c) = exp(lp_t(c)) - exp(logsumexp_{s: l'_s = c}
// (alpha_t(s) + beta_t(s)) + nll - lp_t(c)); zero beyond the input length; reduction 'mean' = mean_b(nll_b / max(L_b, 1)).
#include "xps_common.h"

namespace {
constexpr int CTC_MAXS = 1024;       // 2 * max target length + 1 states held in LDS
constexpr float NEG_INF = -INFINITY;

__device__ inline float lse3(float a, float b, float c) {
    float m = fmaxf(a, fmaxf(b, c));
    if (m == NEG_INF) m = 0.f;
    return logf(expf(a - m) + expf(b - m) + expf(c - m)) + m;
}

__device__ inline int ext_label(const long long* __restrict__ tgt, int s, int blank) {
    return (s & 1) ? (int)tgt[s >> 1] : blank;
}

__global__ __launch_bounds__(64) void ctc_kernel(const float* __restrict__ logits, const long long* __restrict__ targets,
                                                 long long tstride, const long long* __restrict__ in_len,
                                                 const long long* __restrict__ tg_len, int T, int B, int C, int blank,
                                                 int zero_inf, float* __restrict__ nll_out, float* __restrict__ dlogits,
                                                 float* __restrict__ ws_alpha, float* __restrict__ ws_lse, int Smax) {
    __shared__ float row[2][CTC_MAXS + 2];      // recursion rows (index s + 2: two -inf guards in front / behind)
    __shared__ int lab[CTC_MAXS];
    __shared__ float s_nll;
    const int b = blockIdx.x, lane = threadIdx.x;
    const long long* tgt = targets + (long long)b * tstride;
    int Tb = (int)in_len[b];
    Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    int Lb = (int)tg_len[b];
    Lb = Lb < 0 ? 0 : (2 * Lb + 1 > Smax ? (Smax - 1) / 2 : Lb);
    const int S = 2 * Lb + 1;
    float* alpha = ws_alpha + (long long)b * T * Smax;          // [t][s]
    float* lse = ws_lse + (long long)b * T;
    const float* lg = logits + (long long)b * C;                 // element (t, c) at lg[t * B * C + c]
    const long long tB = (long long)B * C;
    for (int s = lane; s < S; s += 64) {
        const int l = ext_label(tgt, s, blank);
        lab[s] = l < 0 ? 0 : (l >= C ? C - 1 : l);          // memory safety only: labels are the caller's contract
    }
    // log-sum-exp of every time step (lanes over t)
    for (int t = lane; t < Tb; t += 64) {
        const float* p = lg + t * tB;
        float mx = p[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(p[c] - mx);
        lse[t] = logf(sum) + mx;
    }
    for (int s = lane; s < CTC_MAXS + 2; s += 64) { row[0][s] = NEG_INF; row[1][s] = NEG_INF; }
    __syncthreads();
    float nll = INFINITY;
    if (Tb > 0) {
        // ---- alpha ----
        for (int s = lane; s < S; s += 64) {
            const float v = s < 2 ? lg[lab[s]] - lse[0] : NEG_INF;
            row[0][s + 2] = v;
            alpha[s] = v;
        }
        __syncthreads();
        for (int t = 1; t < Tb; ++t) {
            const float* prev = row[(t - 1) & 1];
            float* cur = row[t & 1];
            const float* p = lg + t * tB;
            const float l = lse[t];
            for (int s = lane; s < S; s += 64) {
                const float a1 = prev[s + 2], a2 = prev[s + 1];
                const float a3 = (s >= 2 && lab[s] != blank && lab[s] != lab[s - 2]) ? prev[s] : NEG_INF;
                const float v = lse3(a1, a2, a3) + (p[lab[s]] - l);
                cur[s + 2] = v;
                alpha[(long long)t * Smax + s] = v;
            }
            __syncthreads();
        }
        if (lane == 0) {
            const float* last = row[(Tb - 1) & 1];
            const float l1 = last[S - 1 + 2], l2 = S > 1 ? last[S - 2 + 2] : NEG_INF;
            float m = fmaxf(l1, l2);
            if (m == NEG_INF) m = 0.f;
            s_nll = -(logf(expf(l1 - m) + expf(l2 - m)) + m);
        }
        __syncthreads();
        nll = s_nll;
    }
    const bool dead = zero_inf && (nll == INFINITY);
    if (lane == 0) nll_out[b] = dead ? 0.f : nll;
    if (!dlogits) return;
    // gradient of mean_b(nll_b / max(L_b, 1)) with respect to the logits
    float* dl = dlogits + (long long)b * C;
    const float gscale = 1.f / ((float)B * (float)(Lb > 1 ? Lb : 1));
    for (int t = Tb; t < T; ++t)
        for (int c = lane; c < C; c += 64) dl[t * tB + c] = 0.f;
    if (Tb == 0) return;
    if (dead) {
        for (int t = 0; t < Tb; ++t)
            for (int c = lane; c < C; c += 64) dl[t * tB + c] = 0.f;
        return;
    }
    __syncthreads();
    for (int s = lane; s < CTC_MAXS + 2; s += 64) { row[0][s] = NEG_INF; row[1][s] = NEG_INF; }
    __syncthreads();
    // ---- beta, walking backwards; row index s (two -inf guards BEHIND: s + 1, s + 2 may run past S - 1) ----
    for (int t = Tb - 1; t >= 0; --t) {
        const float* nxt = row[(t + 1) & 1];
        float* cur = row[t & 1];
        const float* p = lg + t * tB;
        const float l = lse[t];
        for (int s = lane; s < S; s += 64) {
            float v;
            if (t == Tb - 1) {
                v = (s >= S - 2) ? p[lab[s]] - l : NEG_INF;
            } else {
                const float b1 = nxt[s], b2 = s + 1 < S ? nxt[s + 1] : NEG_INF;
                const float b3 = (s + 2 < S && lab[s + 2] != blank && lab[s + 2] != lab[s]) ? nxt[s + 2] : NEG_INF;
                v = lse3(b1, b2, b3) + (p[lab[s]] - l);
            }
            cur[s] = v;
        }
        __syncthreads();
        // class-wise log-sum-exp of alpha_t(s) + beta_t(s)  (lanes over classes, states walked in order)
        const float* al = alpha + (long long)t * Smax;
        for (int c = lane; c < C; c += 64) {
            float m = NEG_INF;
            for (int s = 0; s < S; ++s)
                if (lab[s] == c) m = fmaxf(m, al[s] + cur[s]);
            float res = NEG_INF;
            if (m != NEG_INF) {
                float sum = 0.f;
                for (int s = 0; s < S; ++s)
                    if (lab[s] == c) sum += expf(al[s] + cur[s] - m);
                res = logf(sum) + m;
            }
            const float lp = p[c] - l;
            dl[t * tB + c] = (expf(lp) - expf(res + nll - lp)) * gscale;
        }
        __syncthreads();
    }
}

// single block: mean over the batch of nll_b / max(L_b, 1), fixed order
__global__ __launch_bounds__(256) void ctc_mean_kernel(const float* __restrict__ nll, const long long* __restrict__ tg_len,
                                                       int B, float* __restrict__ loss) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) {
        const long long L = tg_len[b];
        a += (double)nll[b] / (double)(L > 1 ? L : 1);
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(sh[0] / (double)B);
}
}  // namespace

extern "C" size_t xps_ctc_loss_f32_workspace(int T, int B, int max_target_len) {
    if (T < 1 || B < 1 || max_target_len < 0) return 16;
    const size_t S = 2 * (size_t)max_target_len + 1;
    return ((size_t)T * B * S + (size_t)T * B) * sizeof(float) + 16;
}

extern "C" int xps_ctc_loss_f32(const float* logits, const int64_t* targets, int64_t target_stride,
                                const int64_t* input_lengths, const int64_t* target_lengths, int T, int B, int C,
                                int max_target_len, int blank, int zero_infinity, float* nll, float* loss,
                                float* dlogits, void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(logits && targets && input_lengths && target_lengths && nll && loss, "null argument");
    XPS_CHECK_ARG(T >= 1 && B >= 1 && C >= 1 && max_target_len >= 0 && blank >= 0 && blank < C, "bad size");
    XPS_CHECK_ARG(2 * max_target_len + 1 <= CTC_MAXS, "target longer than 511 labels");
    XPS_CHECK_ARG(target_stride >= max_target_len, "target stride smaller than the longest target");
    if (!workspace || workspace_bytes < xps_ctc_loss_f32_workspace(T, B, max_target_len)) {
        xps_set_error("xps_ctc_loss_f32: workspace too small");
        return XPS_E_WORKSPACE;
    }
    const int Smax = 2 * max_target_len + 1;
    float* ws_alpha = (float*)workspace;
    float* ws_lse = ws_alpha + (size_t)T * B * Smax;
    hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, (const long long*)targets,
                       (long long)target_stride, (const long long*)input_lengths, (const long long*)target_lengths, T, B, C,
                       blank, zero_infinity, nll, dlogits, ws_alpha, ws_lse, Smax);
    XPS_CHECK_LAUNCH();
    hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, nll, (const long long*)target_lengths, B,
                       loss);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}
