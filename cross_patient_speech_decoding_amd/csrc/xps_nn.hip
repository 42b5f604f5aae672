// HBM-bound passes around the MFMA kernels: column reductions (bias gradients,
// BatchNorm statistics), fused BatchNorm(+ReLU)(+dropout) forward/backward, decoder
// glue (row gather / scatter, next-token select), cross-entropy, and the fused
// clip-by-global-norm + AdamW update over the flat parameter buffer.
// All reductions are two-stage and deterministic (no float atomics).
#include "xps_common.h"

namespace {

constexpr int RED_ROWS = 64;    // rows folded by one block of the first reduction stage (16 per thread)

// stage 1: block (bx, by) sums rows [by*RED_ROWS, ...) of columns [bx*64, bx*64+64)
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ X, long long ldx, int rows, int cols,
                                                     float* __restrict__ part, float* __restrict__ part_sq) {
    __shared__ float s1[4][64], s2[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rq = threadIdx.x >> 6;
    const int r0 = blockIdx.y * RED_ROWS;
    const int r1 = min(rows, r0 + RED_ROWS);
    float a = 0.f, a2 = 0.f;
    if (c < cols) {
#pragma unroll 4
        for (int r = r0 + rq; r < r1; r += 4) {
            float v = X[(long long)r * ldx + c];
            a += v;
            a2 += v * v;
        }
    }
    s1[rq][threadIdx.x & 63] = a;
    s2[rq][threadIdx.x & 63] = a2;
    __syncthreads();
    if (rq == 0 && c < cols) {
        const int l = threadIdx.x;
        part[(long long)blockIdx.y * cols + c] = (s1[0][l] + s1[1][l]) + (s1[2][l] + s1[3][l]);
        if (part_sq) part_sq[(long long)blockIdx.y * cols + c] = (s2[0][l] + s2[1][l]) + (s2[2][l] + s2[3][l]);
    }
}

// stage 2: one 1024-thread block per 64 columns; sixteen row-groups fold the partials (8 loads in flight
// each), then combine in a fixed order -- the result depends only on (rows, cols), never on timing
constexpr int S2_GROUPS = 16;
__global__ __launch_bounds__(1024) void colsum_stage2(const float* __restrict__ part, const float* __restrict__ part_sq,
                                                      int nparts, int cols, float* __restrict__ out,
                                                      float* __restrict__ out_sq, int accumulate,
                                                      float* __restrict__ acc_lo = nullptr, float* __restrict__ acc_hi = nullptr,
                                                      int split = 0) {
    __shared__ float s1[S2_GROUPS][64], s2[S2_GROUPS][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + l;
    float a = 0.f, a2 = 0.f;
    if (c < cols) {
        int i = q;
        for (; i + 7 * S2_GROUPS < nparts; i += 8 * S2_GROUPS) {
            float v[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = part[(long long)(i + u * S2_GROUPS) * cols + c];
                w[u] = part_sq ? part_sq[(long long)(i + u * S2_GROUPS) * cols + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += v[u]; a2 += w[u]; }
        }
        for (; i < nparts; i += S2_GROUPS) {
            a += part[(long long)i * cols + c];
            if (part_sq) a2 += part_sq[(long long)i * cols + c];
        }
    }
    s1[q][l] = a;
    s2[q][l] = a2;
    __syncthreads();
    if (q == 0 && c < cols) {
        a = 0.f; a2 = 0.f;
#pragma unroll
        for (int k = 0; k < S2_GROUPS; ++k) { a += s1[k][l]; a2 += s2[k][l]; }
        if (accumulate) {
            a += out[c];
            if (out_sq) a2 += out_sq[c];
        }
        out[c] = a;
        if (out_sq) out_sq[c] = a2;
        // optional: the same sums ADDED into two more vectors (columns < split / >= split): the BatchNorm
        // bias / weight gradients go straight into the flat gradient buffer
        if (acc_lo && c < split) acc_lo[c] += a;
        if (acc_hi && c >= split) acc_hi[c - split] += a;
    }
}

__global__ void bn_finalize_kernel(const float* __restrict__ stats, double count, float* mean, float* rstd,
                                   float* running_mean, float* running_var, long long* num_batches_tracked,
                                   float momentum, float eps, int F) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (c >= F) return;
    const double m = (double)stats[c] / count;
    double var = (double)stats[F + c] / count - m * m;
    if (var < 0) var = 0;
    mean[c] = (float)m;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = count > 1 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

__global__ void bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ rstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ mask, float scale, float* __restrict__ out,
                                long long total, int F, int relu) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % F);
        float v = (y[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
        if (relu) v = v > 0.f ? v : 0.f;
        if (mask) v = v * mask[i] * scale;
        out[i] = v;
    }
}

// finalize + apply in ONE launch: every block derives mean / rstd of all F channels from the (all-reduced) sums into LDS
// with the arithmetic of bn_finalize_kernel (bitwise the same values), block 0 also stores them for the backward pass and
// updates the running statistics; then the grid-stride normalisation of bn_apply_kernel.
__global__ void bn_finalize_apply_kernel(const float* __restrict__ y, const float* __restrict__ stats, double count,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* mean_out, float* rstd_out, float* running_mean, float* running_var,
                                         long long* num_batches_tracked, float momentum, float eps,
                                         const float* __restrict__ mask, float scale, float* __restrict__ out,
                                         long long total, int F, int relu) {
    extern __shared__ float bn_lds[];              // [F] scale = rstd * gamma, [F] shift = beta - mean * rstd * gamma ... kept as mean, rstd
    float* s_mean = bn_lds;
    float* s_rstd = bn_lds + F;
    for (int c = threadIdx.x; c < F; c += blockDim.x) {
        const double m = (double)stats[c] / count;
        double var = (double)stats[F + c] / count - m * m;
        if (var < 0) var = 0;
        const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
        s_mean[c] = mf;
        s_rstd[c] = rf;
        if (blockIdx.x == 0) {
            mean_out[c] = mf;
            rstd_out[c] = rf;
            if (running_mean) {
                const double unbiased = count > 1 ? var * count / (count - 1.0) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % F);
        float v = (y[i] - s_mean[c]) * s_rstd[c] * gamma[c] + beta[c];
        if (relu) v = v > 0.f ? v : 0.f;
        if (mask) v = v * mask[i] * scale;
        out[i] = v;
    }
}

__global__ void bn_apply_eval_kernel(const float* __restrict__ y, const float* __restrict__ rm, const float* __restrict__ rv,
                                     float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float* __restrict__ out, long long total, int F, int relu) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % F);
        float v = (y[i] - rm[c]) * (1.0f / sqrtf(rv[c] + eps)) * gamma[c] + beta[c];
        if (relu) v = v > 0.f ? v : 0.f;
        out[i] = v;
    }
}

// g = dout * mask*scale * relu'(out);  part[by][c] = sum g ; part[by][F + c] = sum g * xhat
__global__ __launch_bounds__(256) void bn_bwd_stage1(const float* __restrict__ dout, const float* __restrict__ out,
                                                     const float* __restrict__ y, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ mask,
                                                     float scale, int relu, long long rows, int F, float* __restrict__ part) {
    __shared__ float s1[4][64], s2[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rq = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.y * RED_ROWS;
    const long long r1 = r0 + RED_ROWS < rows ? r0 + RED_ROWS : rows;
    float a = 0.f, a2 = 0.f;
    if (c < F) {
        const float m = mean[c], rs = rstd[c];
#pragma unroll 4
        for (long long r = r0 + rq; r < r1; r += 4) {
            const long long i = r * F + c;
            float g = dout[i];
            if (mask) g = g * mask[i] * scale;
            if (relu && !(out[i] > 0.f)) g = 0.f;
            a += g;
            a2 += g * ((y[i] - m) * rs);
        }
    }
    s1[rq][threadIdx.x & 63] = a;
    s2[rq][threadIdx.x & 63] = a2;
    __syncthreads();
    if (rq == 0 && c < F) {
        const int l = threadIdx.x;
        part[(long long)blockIdx.y * 2 * F + c] = (s1[0][l] + s1[1][l]) + (s1[2][l] + s1[3][l]);
        part[(long long)blockIdx.y * 2 * F + F + c] = (s2[0][l] + s2[1][l]) + (s2[2][l] + s2[3][l]);
    }
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ y,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ mask, float scale, int relu,
                                    const float* __restrict__ sums, float inv_count, float* __restrict__ dy,
                                    long long total, int F) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % F);
        float g = dout[i];
        if (mask) g = g * mask[i] * scale;
        if (relu && !(out[i] > 0.f)) g = 0.f;
        const float xh = (y[i] - mean[c]) * rstd[c];
        dy[i] = gamma[c] * rstd[c] * (g - sums[c] * inv_count - xh * sums[F + c] * inv_count);
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ table, const long long* __restrict__ idx, float* __restrict__ out,
                                   int B, int cols, int n_rows) {
    const long long total = (long long)B * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / cols), c = (int)(i % cols);
        long long r = idx[b];
        r = r < 0 ? 0 : (r >= n_rows ? n_rows - 1 : r);
        out[i] = table[r * cols + c];
    }
}

// scatter-add of batch rows into a small table, deterministic, two stages:
// stage 1: block (strip of 64 columns, chunk of SC_CHUNK batch rows) accumulates into an LDS copy of the
//          table (each thread owns one (row-group, column): no atomics) and writes its partial table;
// stage 2: partial tables are summed in chunk order.
constexpr int SC_CHUNK = 64, SC_MAXROWS = 16;
__global__ __launch_bounds__(256) void scatter_rows_stage1(const float* __restrict__ dout, const long long* __restrict__ idx,
                                                           float* __restrict__ part, int B, int cols, int n_rows) {
    __shared__ float tab[4][SC_MAXROWS][64];
    const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + l;
    for (int r = 0; r < n_rows; ++r) tab[q][r][l] = 0.f;
    const int b0 = blockIdx.y * SC_CHUNK;
    const int b1 = min(B, b0 + SC_CHUNK);
    if (c < cols)
        for (int b = b0 + q; b < b1; b += 4) {
            long long r = idx[b];
            r = r < 0 ? 0 : (r >= n_rows ? n_rows - 1 : r);
            tab[q][r][l] += dout[(long long)b * cols + c];
        }
    __syncthreads();
    if (c < cols)
        for (int r = q; r < n_rows; r += 4)
            part[((long long)blockIdx.y * n_rows + r) * cols + c] = (tab[0][r][l] + tab[1][r][l]) + (tab[2][r][l] + tab[3][r][l]);
}
// one block = 64 table entries x 4 chunk lanes (a wave each, 8 loads in flight); fixed fold order
__global__ __launch_bounds__(256) void scatter_rows_stage2(const float* __restrict__ part, int nchunks, float* __restrict__ dtable,
                                                           long long total, int accumulate) {
    __shared__ float red[4][64];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + l;
    float a = 0.f;
    if (i < total) {
        int k = w;
        for (; k + 28 < nchunks; k += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(long long)(k + 4 * u) * total + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; k < nchunks; k += 4) a += part[(long long)k * total + i];
    }
    red[w][l] = a;
    __syncthreads();
    if (w == 0 && i < total) {
        a = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
        dtable[i] = accumulate ? dtable[i] + a : a;
    }
}

__global__ void next_token_kernel(const float* __restrict__ logits, int C, const long long* __restrict__ teacher,
                                  long long tstride, const int* __restrict__ use_teacher, long long* __restrict__ next, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (use_teacher && teacher && use_teacher[0]) {
        next[b] = teacher[(long long)b * tstride];
        return;
    }
    const float* p = logits + (long long)b * C;
    float best = p[0];
    int bi = 0;
    for (int c = 1; c < C; ++c)
        if (p[c] > best) { best = p[c]; bi = c; }
    next[b] = bi;
}

// Decode-step glue of the wide decoder (functional.DecoderWideFn): logits = h W_fc^T + b (C <= 16 classes), the next token
// (teacher token if the device flag says so, else the first maximum) and the gather of that token's row of the projection
// table, one wave per trial: lane l holds columns 4 l + 256 j of h; per class a 64-lane butterfly sum in a fixed order.
constexpr int SEL_MAXC = 16;
__global__ __launch_bounds__(256) void decoder_select_kernel(const float* __restrict__ h, const float* __restrict__ w_fc,
                                                             const float* __restrict__ b_fc, float* __restrict__ logits,
                                                             const long long* __restrict__ teacher, long long tstride,
                                                             const int* __restrict__ use_teacher, const float* __restrict__ table,
                                                             long long* __restrict__ next, float* __restrict__ gi_next,
                                                             int B, int H, int C, int ntok) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float acc[SEL_MAXC];
#pragma unroll
    for (int c = 0; c < SEL_MAXC; ++c) acc[c] = 0.f;
    const float* hr = h + (long long)b * H;
    for (int k = lane * 4; k < H; k += 256) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hr + k);
#pragma unroll
        for (int c = 0; c < SEL_MAXC; ++c) {
            if (c < C) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w_fc + (long long)c * H + k);
                acc[c] += (hv[0] * wv[0] + hv[1] * wv[1]) + (hv[2] * wv[2] + hv[3] * wv[3]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < SEL_MAXC; ++c) {
        if (c < C) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
        }
    }
    // every lane now holds the same sums
    float best = 0.f;
    int bi = 0;
#pragma unroll
    for (int c = 0; c < SEL_MAXC; ++c) {
        if (c < C) {
            const float v = acc[c] + b_fc[c];
            if (lane == c) logits[(long long)b * C + c] = v;
            if (c == 0 || v > best) { best = v; bi = c; }
        }
    }
    if (!next) return;
    long long tok = bi;
    if (use_teacher && teacher && use_teacher[0]) tok = teacher[(long long)b * tstride];
    if (lane == 0) next[b] = tok;
    if (gi_next && tok >= 0 && tok < ntok) {
        const float* src = table + tok * 3LL * H;
        float* dst = gi_next + (long long)b * 3 * H;
        for (int k = lane * 4; k < 3 * H; k += 256) *reinterpret_cast<f32x4*>(dst + k) = *reinterpret_cast<const f32x4*>(src + k);
    }
}

__global__ void mask_scale_kernel(const float* __restrict__ x, const float* __restrict__ mask, float scale,
                                  float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = x[i] * mask[i] * scale;
}

// (rng_pair / dropout_keep4: xps_common.h, shared with the GRU kernels that fuse the inter-layer dropout)
// mask[i] = (u_i >= p); optionally out[i] = x[i] * mask[i] * scale in the same pass.  mask may be NULL: the backward
// pass regenerates the decisions from (seed, index) with the same call on the incoming gradient instead of
// reading a stored mask.  VEC: 16-byte accesses, two RNG pairs per thread (n % 4 == 0, 16-byte aligned buffers).
template <bool VEC>
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, float* __restrict__ mask,
                               long long n, float p, float scale, unsigned long long seed) {
    if (VEC) {
        const long long nquad = n / 4;
        for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (long long)gridDim.x * blockDim.x) {
            const f32x4 m = dropout_keep4(seed, q, p);
            if (mask) *reinterpret_cast<f32x4*>(mask + 4 * q) = m;
            if (x) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * q);
                *reinterpret_cast<f32x4*>(out + 4 * q) = v * m * scale;
            }
        }
        return;
    }
    const long long npair = (n + 1) / 2;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < npair; q += (long long)gridDim.x * blockDim.x) {
        float u0, u1;
        rng_pair(seed, q, u0, u1);
        const long long i = 2 * q;
        const float m0 = u0 >= p ? 1.f : 0.f, m1 = u1 >= p ? 1.f : 0.f;
        if (i + 1 < n) {
            if (mask) *reinterpret_cast<float2*>(mask + i) = make_float2(m0, m1);
            if (x) {
                const float2 v = *reinterpret_cast<const float2*>(x + i);
                *reinterpret_cast<float2*>(out + i) = make_float2(v.x * m0 * scale, v.y * m1 * scale);
            }
        } else {
            if (mask) mask[i] = m0;
            if (x) out[i] = x[i] * m0 * scale;
        }
    }
}

// out = XPS_FMT_SPLIT4 image of dropout(x) (p = 0: of x itself): same values as dropout_kernel<true>, then split4_pack
__global__ void split4_kernel(const float* __restrict__ x, float* __restrict__ out, long long nquad, float p, float scale,
                              unsigned long long seed) {
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (long long)gridDim.x * blockDim.x) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * q);
        if (p > 0.f) v = v * dropout_keep4(seed, q, p) * scale;
        *reinterpret_cast<f32x4*>(out + 4 * q) = split4_pack(v);
    }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = a[i] + b[i];
}

__global__ void ce_rows_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                               float* __restrict__ row_loss, long long rows, int C) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* p = logits + r * C;
    float mx = p[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(p[c] - mx);
    row_loss[r] = (logf(s) + mx) - p[target[r]];
}

// single block: ordered tree over row losses -> mean
__global__ __launch_bounds__(1024) void ce_mean_kernel(const float* __restrict__ row_loss, long long rows, float* loss) {
    __shared__ double sh[1024];
    double a = 0.0;
    for (long long i = threadIdx.x; i < rows; i += 1024) a += (double)row_loss[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(sh[0] / (double)rows);
}

// rows <= CE_FUSED_ROWS: per-row losses and their mean in ONE single-block launch (each thread walks rows tid,
// tid + 1024, ...; fixed-order tree).  The two-kernel form costs a second ~5 us launch for 6144 rows of 9 classes.
constexpr int CE_FUSED_ROWS = 1024;      // one row per thread: beyond that the single block is latency-bound (6144 rows: 26 us vs 9.4 us for two launches)
__global__ __launch_bounds__(1024) void ce_fused_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                        float* __restrict__ row_loss, float* __restrict__ loss, long long rows,
                                                        int C) {
    __shared__ double sh[1024];
    double a = 0.0;
    for (long long r = threadIdx.x; r < rows; r += 1024) {
        const float* p = logits + r * C;
        float mx = p[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(p[c] - mx);
        const float l = (logf(s) + mx) - p[target[r]];
        row_loss[r] = l;
        a += (double)l;
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(sh[0] / (double)rows);
}

__global__ void ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                              const float* __restrict__ gout, float* __restrict__ dlogits, long long rows, int C) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* p = logits + r * C;
    float mx = p[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(p[c] - mx);
    const float g = gout[0] / (float)rows, inv = 1.f / s;
    const long long tg = target[r];
    for (int c = 0; c < C; ++c) dlogits[r * C + c] = g * (expf(p[c] - mx) * inv - (c == tg ? 1.f : 0.f));
}

constexpr int SS_CHUNK = 2048;
__global__ __launch_bounds__(256) void sumsq_stage1(const float* __restrict__ g, long long n, double* __restrict__ part) {
    __shared__ double sh[256];
    const long long i0 = (long long)blockIdx.x * SS_CHUNK;
    const long long i1 = i0 + SS_CHUNK < n ? i0 + SS_CHUNK : n;
    double a = 0.0;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) { const double v = g[i]; a += v * v; }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void sumsq_stage2(const double* __restrict__ part, int nparts, float* sumsq) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) sumsq[0] = (float)sh[0];
}

__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n, const float* __restrict__ sumsq, float max_norm, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2_sqrt) {
    float coef = 1.f;
    if (max_norm > 0.f && sumsq) {
        const float c = max_norm / (sqrtf(sumsq[0]) + 1e-6f);
        coef = c < 1.f ? c : 1.f;
    }
    const float step_size = lr / bc1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        g[i] = gi;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

// clip + AdamW with the second stage of the gradient-norm reduction folded in: every block folds the (few hundred)
// partial sums in the same fixed order, block 0 publishes the total
__global__ __launch_bounds__(256) void adamw_fused_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, long long n, const double* __restrict__ part,
                                                          int nparts, float* __restrict__ sumsq_out, float max_norm, float lr,
                                                          float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    __shared__ double sh[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    const float ss = (float)sh[0];
    if (blockIdx.x == 0 && threadIdx.x == 0 && sumsq_out) sumsq_out[0] = ss;
    float coef = 1.f;
    if (max_norm > 0.f) {
        const float c = max_norm / (sqrtf(ss) + 1e-6f);
        coef = c < 1.f ? c : 1.f;
    }
    const float step_size = lr / bc1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        g[i] = gi;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

inline int ew_grid(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" size_t xps_colsum_f32_workspace(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 16;
    return (size_t)2 * cdiv(rows, RED_ROWS) * cols * sizeof(float) + 16;
}

extern "C" int xps_colsum_f32(const float* X, int64_t ldx, int rows, int cols, float* out, float* out_sq,
                              int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(X && out && rows >= 0 && cols >= 1, "bad argument");
    if (workspace_bytes < xps_colsum_f32_workspace(rows, cols) || !workspace) {
        xps_set_error("xps_colsum_f32: workspace too small");
        return XPS_E_WORKSPACE;
    }
    const int nparts = rows > 0 ? cdiv(rows, RED_ROWS) : 0;
    float* part = (float*)workspace;
    float* part_sq = out_sq ? part + (size_t)nparts * cols : nullptr;
    if (nparts > 0) {
        hipLaunchKernelGGL(colsum_stage1, dim3(cdiv(cols, 64), nparts), dim3(256), 0, (hipStream_t)stream,
                           X, (long long)ldx, rows, cols, part, part_sq);
        XPS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(colsum_stage2, dim3(cdiv(cols, 64)), dim3(1024), 0, (hipStream_t)stream,
                       part, part_sq, nparts, cols, out, out_sq, accumulate);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_bn_finalize_f32(const float* stats, double count, float* mean, float* rstd,
                                   float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                   float momentum, float eps, int F, void* stream) {
    XPS_CHECK_ARG(stats && mean && rstd && F >= 1 && count >= 1, "bad argument");
    XPS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running stats must both be given or both NULL");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(F, 256)), dim3(256), 0, (hipStream_t)stream,
                       stats, count, mean, rstd, running_mean, running_var, (long long*)num_batches_tracked, momentum, eps, F);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_bn_apply_f32(const float* y, const float* mean, const float* rstd, const float* gamma,
                                const float* beta, const float* drop_mask, float drop_scale, float* out,
                                int64_t rows, int F, int relu, void* stream) {
    XPS_CHECK_ARG(y && mean && rstd && gamma && beta && out && rows >= 0 && F >= 1, "bad argument");
    const long long total = (long long)rows * F;
    if (total == 0) return XPS_OK;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       y, mean, rstd, gamma, beta, drop_mask, drop_scale, out, total, F, relu);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_bn_finalize_apply_f32(const float* y, const float* stats, double count, const float* gamma, const float* beta,
                                         float* mean, float* rstd, float* running_mean, float* running_var,
                                         int64_t* num_batches_tracked, float momentum, float eps, const float* drop_mask,
                                         float drop_scale, float* out, int64_t rows, int F, int relu, void* stream) {
    XPS_CHECK_ARG(y && stats && gamma && beta && mean && rstd && out && rows >= 1 && F >= 1 && F <= 8192 && count > 0, "bad argument");
    XPS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running_mean and running_var go together");
    const long long total = (long long)rows * F;
    hipLaunchKernelGGL(bn_finalize_apply_kernel, dim3(ew_grid(total)), dim3(256), 2 * F * sizeof(float), (hipStream_t)stream,
                       y, stats, count, gamma, beta, mean, rstd, running_mean, running_var, (long long*)num_batches_tracked,
                       momentum, eps, drop_mask, drop_scale, out, total, F, relu);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_bn_apply_eval_f32(const float* y, const float* running_mean, const float* running_var, float eps,
                                     const float* gamma, const float* beta, float* out, int64_t rows, int F,
                                     int relu, void* stream) {
    XPS_CHECK_ARG(y && running_mean && running_var && gamma && beta && out && rows >= 0 && F >= 1, "bad argument");
    const long long total = (long long)rows * F;
    if (total == 0) return XPS_OK;
    hipLaunchKernelGGL(bn_apply_eval_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       y, running_mean, running_var, eps, gamma, beta, out, total, F, relu);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" size_t xps_bn_bwd_workspace(int64_t rows, int F) {
    if (rows <= 0 || F <= 0) return 16;
    return (size_t)cdiv(rows, RED_ROWS) * 2 * F * sizeof(float) + 16;
}

extern "C" int xps_bn_bwd_reduce_f32(const float* dout, const float* out, const float* y, const float* mean,
                                     const float* rstd, const float* drop_mask, float drop_scale, int relu,
                                     float* sums, float* dbeta_acc, float* dgamma_acc, int64_t rows, int F,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(dout && y && mean && rstd && sums && rows >= 1 && F >= 1, "bad argument");
    XPS_CHECK_ARG(!relu || out, "relu backward needs the forward output");
    if (workspace_bytes < xps_bn_bwd_workspace(rows, F) || !workspace) {
        xps_set_error("xps_bn_bwd_reduce_f32: workspace too small");
        return XPS_E_WORKSPACE;
    }
    const int nparts = cdiv(rows, RED_ROWS);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(bn_bwd_stage1, dim3(cdiv(F, 64), nparts), dim3(256), 0, (hipStream_t)stream,
                       dout, out, y, mean, rstd, drop_mask, drop_scale, relu, (long long)rows, F, part);
    XPS_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_stage2, dim3(cdiv(2 * F, 64)), dim3(1024), 0, (hipStream_t)stream,
                       part, (const float*)nullptr, nparts, 2 * F, sums, (float*)nullptr, 0, dbeta_acc, dgamma_acc, F);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_bn_bwd_apply_f32(const float* dout, const float* out, const float* y, const float* mean,
                                    const float* rstd, const float* gamma, const float* drop_mask, float drop_scale,
                                    int relu, const float* sums, double count, float* dy, int64_t rows, int F,
                                    void* stream) {
    XPS_CHECK_ARG(dout && y && mean && rstd && gamma && sums && dy && rows >= 0 && F >= 1 && count >= 1, "bad argument");
    XPS_CHECK_ARG(!relu || out, "relu backward needs the forward output");
    const long long total = (long long)rows * F;
    if (total == 0) return XPS_OK;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                       dout, out, y, mean, rstd, gamma, drop_mask, drop_scale, relu, sums, (float)(1.0 / count), dy,
                       total, F);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_gather_rows_f32(const float* table, const int64_t* idx, float* out, int B, int cols, int n_rows,
                                   void* stream) {
    XPS_CHECK_ARG(table && idx && out && B >= 0 && cols >= 1 && n_rows >= 1, "bad argument");
    if (B == 0) return XPS_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(ew_grid((long long)B * cols)), dim3(256), 0, (hipStream_t)stream,
                       table, (const long long*)idx, out, B, cols, n_rows);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" size_t xps_scatter_rows_f32_workspace(int B, int cols, int n_rows) {
    if (B <= 0 || cols <= 0 || n_rows <= 0) return 16;
    return (size_t)cdiv(B, SC_CHUNK) * n_rows * cols * sizeof(float) + 16;
}

extern "C" int xps_scatter_rows_f32(const float* dout, const int64_t* idx, float* dtable, int B, int cols, int n_rows,
                                    int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(dout && idx && dtable && B >= 0 && cols >= 1 && n_rows >= 1, "bad argument");
    XPS_CHECK_ARG(n_rows <= SC_MAXROWS, "table has too many rows for the scatter kernel (max 16)");
    if (workspace_bytes < xps_scatter_rows_f32_workspace(B, cols, n_rows) || !workspace) {
        xps_set_error("xps_scatter_rows_f32: workspace too small");
        return XPS_E_WORKSPACE;
    }
    const int nchunks = B > 0 ? cdiv(B, SC_CHUNK) : 0;
    float* part = (float*)workspace;
    if (nchunks > 0) {
        hipLaunchKernelGGL(scatter_rows_stage1, dim3(cdiv(cols, 64), nchunks), dim3(256), 0, (hipStream_t)stream,
                           dout, (const long long*)idx, part, B, cols, n_rows);
        XPS_CHECK_LAUNCH();
    }
    const long long total = (long long)n_rows * cols;
    hipLaunchKernelGGL(scatter_rows_stage2, dim3(cdiv(total, 64)), dim3(256), 0, (hipStream_t)stream,
                       part, nchunks, dtable, total, accumulate);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_next_token(const float* logits, int n_classes, const int64_t* teacher, int64_t teacher_stride,
                              const int32_t* use_teacher, int64_t* next, int B, void* stream) {
    XPS_CHECK_ARG(logits && next && n_classes >= 1 && B >= 0, "bad argument");
    if (B == 0) return XPS_OK;
    hipLaunchKernelGGL(next_token_kernel, dim3(cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, n_classes, (const long long*)teacher, (long long)teacher_stride, (const int*)use_teacher,
                       (long long*)next, B);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_decoder_select_f32(const float* h, const float* w_fc, const float* b_fc, float* logits,
                                      const int64_t* teacher, int64_t teacher_stride, const int32_t* use_teacher,
                                      const float* table, int64_t* next, float* gi_next, int B, int H, int C, int ntok,
                                      void* stream) {
    XPS_CHECK_ARG(h && w_fc && b_fc && logits && B >= 0 && H >= 4 && C >= 1, "bad argument");
    XPS_CHECK_ARG(C <= SEL_MAXC && H % 4 == 0, "at most 16 classes, H a multiple of 4");
    XPS_CHECK_ARG(!gi_next || (table && next && ntok >= 1), "the gather needs the table and the token output");
    const uintptr_t bits = reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(w_fc) | reinterpret_cast<uintptr_t>(table) |
                           reinterpret_cast<uintptr_t>(gi_next);
    XPS_CHECK_ARG((bits & 15) == 0, "h, w_fc, table and gi_next must be 16-byte aligned");
    if (B == 0) return XPS_OK;
    hipLaunchKernelGGL(decoder_select_kernel, dim3(cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, h, w_fc, b_fc, logits,
                       (const long long*)teacher, (long long)teacher_stride, (const int*)use_teacher, table, (long long*)next,
                       gi_next, B, H, C, ntok);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_mask_scale_f32(const float* x, const float* mask, float scale, float* out, int64_t n, void* stream) {
    XPS_CHECK_ARG(x && mask && out && n >= 0, "bad argument");
    if (n == 0) return XPS_OK;
    hipLaunchKernelGGL(mask_scale_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, mask, scale, out,
                       (long long)n);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_dropout_f32(const float* x, float* out, float* mask, int64_t n, float p, uint64_t seed, void* stream) {
    XPS_CHECK_ARG(n >= 0 && p >= 0.f && p < 1.f, "bad argument");
    XPS_CHECK_ARG(mask || x, "nothing to produce: give mask, or x and out");
    XPS_CHECK_ARG((x == nullptr) == (out == nullptr), "x and out must both be given or both NULL");
    const uintptr_t bits = reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out);
    XPS_CHECK_ARG((bits & 7) == 0, "buffers must be 8-byte aligned");
    if (n == 0) return XPS_OK;
    if ((bits & 15) == 0 && n % 4 == 0)
        hipLaunchKernelGGL(dropout_kernel<true>, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, out, mask,
                           (long long)n, p, 1.0f / (1.0f - p), (unsigned long long)seed);
    else
        hipLaunchKernelGGL(dropout_kernel<false>, dim3(ew_grid((n + 1) / 2)), dim3(256), 0, (hipStream_t)stream, x, out, mask,
                           (long long)n, p, 1.0f / (1.0f - p), (unsigned long long)seed);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_split4_f32(const float* x, float* out, int64_t n, float drop_p, uint64_t seed, void* stream) {
    XPS_CHECK_ARG(x && out && n >= 0 && n % 4 == 0, "bad argument (n must be a multiple of 4)");
    XPS_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "drop_p must be in [0, 1)");
    XPS_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "buffers must be 16-byte aligned");
    if (n == 0) return XPS_OK;
    hipLaunchKernelGGL(split4_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, out, (long long)(n / 4), drop_p,
                       1.0f / (1.0f - drop_p), (unsigned long long)seed);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

namespace {
// out[r][0 .. ldo) = the XPS_FMT_SPLIT4 image of x[r][0 .. cols), zero beyond (cols, ldo multiples of 4)
__global__ __launch_bounds__(256) void split4_pad_kernel(const float* __restrict__ x, long long ldx, int rows, int cols, float* __restrict__ out,
                                                         long long ldo) {
    const long long g4 = ldo / 4, idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)rows * g4) return;
    const long long r = idx / g4;
    const int c = (int)(idx % g4) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (c < cols) v = split4_pack(*reinterpret_cast<const f32x4*>(x + r * ldx + c));
    *reinterpret_cast<f32x4*>(out + r * ldo + c) = v;
}
}  // namespace

extern "C" int xps_split4_pad_f32(const float* x, int64_t ldx, int rows, int cols, float* out, int64_t ldo, void* stream) {
    XPS_CHECK_ARG(x && out && rows >= 0 && cols >= 0 && cols % 4 == 0 && ldo % 4 == 0 && ldo >= cols && ldx % 4 == 0 && ldx >= cols, "bad argument");
    XPS_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0, "buffers must be 16-byte aligned");
    if (rows == 0 || ldo == 0) return XPS_OK;
    hipLaunchKernelGGL(split4_pad_kernel, dim3(cdiv((long long)rows * (ldo / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, rows,
                       cols, out, (long long)ldo);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
    XPS_CHECK_ARG(a && b && out && n >= 0, "bad argument");
    if (n == 0) return XPS_OK;
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long long)n);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_cross_entropy_fwd_f32(const float* logits, const int64_t* target, float* row_loss, float* loss,
                                         int64_t rows, int n_classes, void* stream) {
    XPS_CHECK_ARG(logits && target && row_loss && loss && rows >= 1 && n_classes >= 1, "bad argument");
    if (rows <= CE_FUSED_ROWS) {
        hipLaunchKernelGGL(ce_fused_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, (const long long*)target,
                           row_loss, loss, (long long)rows, n_classes);
        XPS_CHECK_LAUNCH();
        return XPS_OK;
    }
    hipLaunchKernelGGL(ce_rows_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, (const long long*)target, row_loss, (long long)rows, n_classes);
    XPS_CHECK_LAUNCH();
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_loss, (long long)rows, loss);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

// Loss AND unit gradient in ONE launch (the three-kernel form costs two more ~5 us launches plus their host time per
// step): every block handles 256 rows -- row losses, d(mean loss)/d(logits) = (softmax - onehot) / rows -- and leaves its
// partial loss sum (double); the block that takes the last ticket adds the partials IN INDEX ORDER (the result does not depend
// on which block that is) and resets the ticket for the next call.  Partials are published with agent-scope fences around
// the ticket (cdna_hip_programming.md, Guideline 16): release before the atomic, acquire after it in the last block.
__global__ __launch_bounds__(256) void ce_loss_grad_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                           float* __restrict__ row_loss, float* __restrict__ loss,
                                                           float* __restrict__ dlogits, double* part, unsigned* ticket,
                                                           long long rows, int C) {
    __shared__ double sh[256];
    __shared__ unsigned last;
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    double l = 0.0;
    if (r < rows) {
        const float* p = logits + r * C;
        float mx = p[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(p[c] - mx);
        const long long tg = target[r];
        const float rl = (logf(s) + mx) - p[tg];
        row_loss[r] = rl;
        l = (double)rl;
        if (dlogits) {
            const float g = 1.f / (float)rows, inv = 1.f / s;
            for (int c = 0; c < C; ++c) dlogits[r * C + c] = g * (expf(p[c] - mx) * inv - (c == tg ? 1.f : 0.f));
        }
    }
    sh[threadIdx.x] = l;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[blockIdx.x] = sh[0];
        __threadfence();                                          // release: the partial before the ticket
        last = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();                                          // acquire: every block's partial
        double a = 0.0;
        for (unsigned b = 0; b < gridDim.x; ++b) a += __hip_atomic_load(part + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        loss[0] = (float)(a / (double)rows);
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call on this stream
    }
}

extern "C" size_t xps_cross_entropy_loss_grad_f32_workspace(int64_t rows) {
    return (size_t)(cdiv(rows > 0 ? rows : 1, 256) + 2) * sizeof(double);
}

extern "C" int xps_cross_entropy_loss_grad_f32(const float* logits, const int64_t* target, float* row_loss, float* loss,
                                               float* dlogits, void* workspace, size_t workspace_bytes, int64_t rows,
                                               int n_classes, void* stream) {
    XPS_CHECK_ARG(logits && target && row_loss && loss && rows >= 1 && n_classes >= 1, "bad argument");
    if (!workspace || workspace_bytes < xps_cross_entropy_loss_grad_f32_workspace(rows) || (reinterpret_cast<uintptr_t>(workspace) & 7)) {
        xps_set_error("xps_cross_entropy_loss_grad_f32: workspace too small or misaligned");
        return XPS_E_WORKSPACE;
    }
    const int blocks = cdiv(rows, 256);
    unsigned* ticket = (unsigned*)workspace;              // FIRST word: its place must not depend on the row count (calls of
    double* part = (double*)workspace + 2;                // different sizes share one zero-initialised buffer)
    hipLaunchKernelGGL(ce_loss_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, (const long long*)target,
                       row_loss, loss, dlogits, part, ticket, (long long)rows, n_classes);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_cross_entropy_bwd_f32(const float* logits, const int64_t* target, const float* gout, float* dlogits,
                                         int64_t rows, int n_classes, void* stream) {
    XPS_CHECK_ARG(logits && target && gout && dlogits && rows >= 1 && n_classes >= 1, "bad argument");
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, (const long long*)target, gout, dlogits, (long long)rows, n_classes);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" size_t xps_sumsq_f32_workspace(int64_t n) {
    return (size_t)(cdiv(n > 0 ? n : 1, SS_CHUNK)) * sizeof(double) + 16;
}

extern "C" int xps_sumsq_f32(const float* g, int64_t n, float* sumsq, void* workspace, size_t workspace_bytes,
                             void* stream) {
    XPS_CHECK_ARG(g && sumsq && n >= 0, "bad argument");
    if (workspace_bytes < xps_sumsq_f32_workspace(n) || !workspace) {
        xps_set_error("xps_sumsq_f32: workspace too small");
        return XPS_E_WORKSPACE;
    }
    XPS_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, "workspace must be 8-byte aligned");
    const int nparts = n > 0 ? cdiv(n, SS_CHUNK) : 0;
    if (nparts > 0) {
        hipLaunchKernelGGL(sumsq_stage1, dim3(nparts), dim3(256), 0, (hipStream_t)stream, g, (long long)n, (double*)workspace);
        XPS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)workspace, nparts, sumsq);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_adamw_f32(float* p, float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
    XPS_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1, "bad argument");
    XPS_CHECK_ARG(max_norm <= 0.f || sumsq, "clipping needs the sum of squares");
    if (n == 0) return XPS_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, sumsq,
                       max_norm, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2));
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_clip_adamw_f32(float* p, float* g, float* m, float* v, int64_t n, float* sumsq, float max_norm, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, int step, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1, "bad argument");
    if (n == 0) return XPS_OK;
    if (!workspace || workspace_bytes < xps_sumsq_f32_workspace(n)) {
        xps_set_error("xps_clip_adamw_f32: workspace too small");
        return XPS_E_WORKSPACE;
    }
    const int nparts = cdiv(n, SS_CHUNK);
    hipLaunchKernelGGL(sumsq_stage1, dim3(nparts), dim3(256), 0, (hipStream_t)stream, g, (long long)n, (double*)workspace);
    XPS_CHECK_LAUNCH();
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_fused_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n,
                       (const double*)workspace, nparts, sumsq, max_norm, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                       (float)sqrt(bc2));
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}
