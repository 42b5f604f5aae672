/* xps.h -- C ABI of libxps.so, the MI355X (gfx950) hot path of the cross-patient
 * speech-decoding trainer.
 *
 * The reference (coganlab/cross_patient_speech_decoding) is pure Python and has no
 * FFI of its own; every entry point below replaces a library call the reference
 * makes on its aligned-training path.  The "replaces" notes cite
 * /root/reference/aligned_decoding/<file>:<line>.
 *
 * Contract (boundary B2)
 *   - plain pointers and sizes only; all data pointers are DEVICE pointers unless a
 *     parameter is documented as host;
 *   - the caller owns every buffer, including workspaces whose size is returned by
 *     the matching xps_*_workspace() query; the library never allocates, frees or
 *     keeps a pointer after the call returns;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); it
 *     never synchronises, so calls compose with autograd ordering and can be
 *     captured into a hipGraph;
 *   - return value: 0 = ok, negative = error (XPS_E_*); xps_last_error() returns a
 *     thread-local message.  No C++ exception crosses the boundary;
 *   - re-entrant.  Global state: exactly two process-wide mode switches, both plain ints read at call time --
 *     xps_set_gemm_precision (product precision of the matrix kernels) and xps_set_gru_cluster_mode (launch form of the
 *     H > 256 recurrence) -- plus a per-device cache of immutable device facts (CU count, resident workgroups of the cluster kernels) and the
 *     caller's per-device status word (xps_gru_set_status_word); nothing else outlives a call.
 *
 * Matrices are row-major float32 unless stated.  A "row map" (rpg, gs, ld) addresses
 * row i of a matrix at element offset  (i / rpg) * gs + (i % rpg) * ld ; an ordinary
 * matrix has rpg >= rows and ld = leading dimension.  It lets one GEMM read the
 * strided-convolution windows of a (trial x time x channel) tensor and write
 * time-major results without a copy.
 */
#ifndef XPS_H
#define XPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XPS_OK 0
#define XPS_E_INVALID (-1)     /* bad argument / unsupported shape */
#define XPS_E_HIP (-2)         /* a HIP runtime call failed */
#define XPS_E_WORKSPACE (-3)   /* workspace too small */

typedef struct xps_rowmap {
    int64_t gs;    /* stride between groups of rows            */
    int64_t ld;    /* stride between rows inside a group       */
    int32_t rpg;   /* rows per group (>= 1)                    */
    int32_t fmt;   /* XPS_FMT_F32 (0) or XPS_FMT_SPLIT4 (1): element format of a GEMM INPUT operand (ignored for outputs) */
} xps_rowmap;
/* XPS_FMT_SPLIT4: every aligned group of four consecutive fp32 elements (16 bytes, along the contiguous index) holds the
 * bf16 split of its values instead: bytes 0-7 = hi[0..3] = bf16(x[0..3]), bytes 8-15 = lo[0..3] = bf16(x[j] - hi[j]) -- exactly
 * what the bf16x3 tile kernels compute while they stage an fp32 operand.  A producer that owns the split anyway (the BPTT
 * kernels through xps_gru_seq_bwd_split4_f32, a dropout pass or a weight matrix through xps_split4_f32) writes it once; the GEMM entry points (nt / nn / nn2 / nt_multi /
 * tn_grouped, bf16x3 mode only) then stage the operand without any conversion arithmetic and return the SAME BITS as for
 * the fp32 operand (a bias gradient folded from a split4 A operand sums hi + lo: within 2^-17 relative of the fp32 sum).
 * Needs 16-byte aligned operands, leading dimensions and the contiguous extent multiples of 4; else XPS_E_INVALID. */
#define XPS_FMT_F32 0
#define XPS_FMT_SPLIT4 1

const char* xps_last_error(void);
int xps_abi_version(void);
/* a HIP stream at the lowest priority of the device (for work that must yield to the caller's main stream) */
int xps_stream_create_low_priority(void** stream);
int xps_stream_destroy(void* stream);
/* Product precision of the tiled GEMM entry points below (xps_gemm_*) and of the fused GRU recurrence (xps_gru_seq_*,
 * every H): 0 = fp32 MFMA, exact fp32 fma chains; 1 = bf16 split products (the default): every fp32 operand is split hi + lo in bf16 while it is staged and a product is
 * accumulated in fp32 as lo*hi + hi*lo + hi*hi on the bf16 matrix pipe (relative product error ~2^-16; results
 * stay inside the 1e-4 parity bar; BASELINE.json names bf16 for this path).  Process-wide, initial value from
 * XPS_GEMM_PRECISION=fp32|bf16x3; returns XPS_E_INVALID for other modes. */
int xps_set_gemm_precision(int mode);
int xps_get_gemm_precision(void);
/* Tile shape of the matrix kernels in bf16x3 mode: 1 (default; XPS_GEMM_BIG=0 in the environment starts with 0) lets large
 * interior shapes (M, N multiples of 256, k ranges multiples of 16, plain 16-byte aligned operands, >= 192 tiles) run on
 * 256 x 256 tiles / 8 waves; 0 keeps every product on the 128 x 128 (64 x 128) tiles.  Same arithmetic per k-tile in both:
 * products whose k range is not split (nt / nn / nn2 / nt_multi) have the same bits either way. */
int xps_set_gemm_big_tiles(int on);
int xps_get_gemm_big_tiles(void);

/* ------------------------------------------------------------------------- */
/* Dense fp32 contractions: bf16 split products (default) or the f32-input     */
/* MFMA with exact fp32 fma chains (xps_set_gemm_precision above).              */
/* Replaces the GEMMs inside torch.nn.Conv1d / nn.GRU / nn.Linear and their     */
/* autograd (nn_models/models.py:616,661,739,746).                               */
/* ------------------------------------------------------------------------- */

/* C[m][n] (+)= sum_k A[m][k] * B[n][k] + bias[n]      (A: M x K, B: N x K)   */
int xps_gemm_nt_f32(const float* A, const xps_rowmap* ra, const float* B, const xps_rowmap* rb,
                    float* C, const xps_rowmap* rc, const float* bias,
                    int M, int N, int K, int accumulate, void* stream);
/* C[m][n] (+)= sum_k A[m][k] * B[k][n]                (A: M x K, B: K x N)   */
int xps_gemm_nn_f32(const float* A, const xps_rowmap* ra, const float* B, const xps_rowmap* rb,
                    float* C, const xps_rowmap* rc,
                    int M, int N, int K, int accumulate, void* stream);
/* nprob (1..4) problems C_i = A B_i^T + bias_i sharing A and all sizes / row maps, in ONE launch (the input
 * projections of every direction of a GRU layer).  B, C, bias are HOST arrays of device pointers.       */
int xps_gemm_nt_multi_f32(const float* A, const xps_rowmap* ra, const float* const* B, const xps_rowmap* rb,
                          float* const* C, const xps_rowmap* rc, const float* const* bias, int nprob,
                          int M, int N, int K, void* stream);
/* C (+)= A1 B1 + A2 B2  (A_i: M x K_i, B_i: K_i x N, both pairs share the row maps): the input gradient of
 * a bidirectional layer, both directions summed in registers in one launch.    */
int xps_gemm_nn2_f32(const float* A1, const float* B1, int K1, const float* A2, const float* B2, int K2,
                     const xps_rowmap* ra, const xps_rowmap* rb, float* C, const xps_rowmap* rc,
                     int M, int N, int accumulate, void* stream);
/* C[m][n] (+)= sum_k A[k][m] * B[k][n]                (A: K x M, B: K x N)
 * split-K over deterministic partial slabs in `workspace`.                    */
size_t xps_gemm_tn_f32_workspace(int M, int N, int K);
int xps_gemm_tn_f32(const float* A, const xps_rowmap* ra, const float* B, const xps_rowmap* rb,
                    float* C, const xps_rowmap* rc,
                    int M, int N, int K, int accumulate,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Grouped form: up to 12 weight-gradient problems  C_p (+)= A_p^T B_p  in ONE launch (+ one
 * reduce launch).  colsum_a, when not NULL, receives  colsum_a[m] (+)= sum_k A_p[k][m]  (the bias
 * gradient) folded from the staged A tiles inside the same launch — no separate reduction pass.
 * `probs` is a HOST array; it is copied into the kernel arguments.                        */
typedef struct xps_tn_problem {
    const float* A; const float* B; float* C; float* colsum_a;
    xps_rowmap ra, rb, rc;
    int32_t M, N, K, accumulate;      /* bit 0: C and colsum_a accumulate; bit 1: colsum_a alone accumulates */
} xps_tn_problem;
size_t xps_gemm_tn_grouped_f32_workspace(const xps_tn_problem* probs, int n);
int xps_gemm_tn_grouped_f32(const xps_tn_problem* probs, int n, void* workspace, size_t workspace_bytes,
                            void* stream);

/* out[c] (+)= sum_r X[r][c] (and optionally sum_r X[r][c]^2 into out_sq), two-stage,
 * deterministic.  Bias gradients and BatchNorm batch statistics.               */
size_t xps_colsum_f32_workspace(int rows, int cols);
int xps_colsum_f32(const float* X, int64_t ldx, int rows, int cols, float* out, float* out_sq,
                   int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Fused GRU recurrence.  Replaces torch.nn.GRU's per-step cell                 */
/* (nn_models/models.py:661-663,687 encoder; :739-740,759 decoder).             */
/*                                                                             */
/* Layout (time-major, all directions in one call):                            */
/*   gi    [ndir][T][B][3H]  x_t W_ih^T + b_ih, gate order r,z,n (PyTorch)      */
/*   w_hh  [ndir] pointers to (3H x H) ; b_hh [ndir] pointers to (3H)           */
/*   h0    [ndir][B][H] or NULL (zeros)                                         */
/*   y_ext [T+2][B][ndir*H]  slot t+1 holds h_t of every direction in ACTUAL     */
/*          time; slot 0 / slot T+1 hold h0 of the forward / reverse direction;  */
/*          direction 1 runs t = T-1 .. 0                                       */
/*   saved ndir*T*B*4H floats: r, z, n, (W_hn h + b_hn) for the backward (or NULL).  OPAQUE: written by the forward entry point and
 *          read by the backward entry point of the same shape and mode only; [ndir][T][B][4H] on most paths, member-major
 *          ([ndir][T][H/32][B][4][32]) on the cluster path when H % 32 == 0 (sequential HBM streams per workgroup)            */
/* ------------------------------------------------------------------------- */
/* 128 < H <= 512 (H % 4 == 0, B >= 128): cluster-persistent recurrence (csrc/xps_gru_cluster.hip): W_hh stays in the registers
 * of a cluster of workgroups for the whole sequence, the members exchange h_t through `workspace` inside ONE launch.
 * xps_set_gru_cluster_mode / XPS_GRU_CLUSTER = off | steps | persistent (0 | 1 | 2, default 2): 1 runs the same kernels one
 * step per launch, 0 the per-step GEMM kernels.  A hand-off that timed out (3 s; e.g. two such launches of different processes
 * sharing one GPU) sets the 32-bit word at byte xps_gru_seq_status_offset() of the workspace to 1 (-1: the shape has no
 * status word); the caller checks it whenever it synchronises anyway.                                                    */
size_t xps_gru_seq_fwd_f32_workspace(int T, int B, int H, int ndir);
long long xps_gru_seq_status_offset(int T, int B, int H, int ndir);
/* A caller-owned, zero-initialised 32-bit DEVICE word for the current device (NULL: none) that outlives the workspaces: a
 * hand-off that timed out stores 1 there as well, so a trainer checks ONE word per device where it reads the loss instead of
 * keeping every launch's workspace alive.  The persistent form is only launched when the occupancy query says the device
 * holds the whole grid at once; otherwise (and always with mode 1) the same kernels run one step per launch.               */
int xps_gru_set_status_word(unsigned* device_word);
/* BPTT kernel of the cluster path in bf16x3 mode, 384 < H <= 512: 0 (default) = every member of a 16-workgroup cluster contracts
 * all 3H gate-gradient columns of its trials; 1 (XPS_GRU_CL2=1) = the members form a 4 x 4 grid, contract one K slice each and
 * exchange partial sums (a third of the LDS-DMA ingest, a second hand-off per step; same results up to summation order;
 * measured on a par: DESIGN.md 4.5).  Process-wide, read at call time; the workspace query covers both.                       */
int xps_set_gru_bptt_grid(int two_dimensional);
int xps_get_gru_bptt_grid(void);
int xps_set_gru_cluster_mode(int mode);
int xps_get_gru_cluster_mode(void);
int xps_gru_seq_fwd_f32(const float* gi, const float* const* w_hh, const float* const* b_hh,
                        const float* h0, float* y_ext, float* saved,
                        int T, int B, int H, int ndir, void* workspace, size_t workspace_bytes, void* stream);

/* Backward through the recurrence (BPTT).
 *   dy     [T][B][ndir*H]   gradient w.r.t. the layer output (actual time), or NULL (all zero)
 *   dhn    [ndir][B][H]     gradient w.r.t. the FINAL hidden state of each direction (h at t = T-1 forward,
 *                           t = 0 reverse), or NULL; it seeds the running dh, so a layer whose per-step
 *                           output is unused downstream needs no dy buffer at all
 *   w_hh   [ndir] pointers to W_hh (3H x H) and  w_hh_t [ndir] pointers to W_hh^T (H x 3H, see
 *          xps_transpose_f32): H <= 128 keeps W_hh^T resident in registers for all steps; larger H runs one
 *          fused GEMM + gate-gradient launch per step on W_hh (workspace: xps_gru_seq_bwd_f32_workspace)
 *   dgi    [ndir][T][B][3H] gradient w.r.t. gi   (= w.r.t. input pre-activations r, z, n)
 *   dghn   [ndir][T][B][H]  n-gate part of the gradient w.r.t. (h_{t-1} W_hh^T + b_hh); its r and z
 *                           parts equal those of dgi, so they are not stored twice
 *   dh0    [ndir][B][H]     gradient w.r.t. h0 (or NULL)                       */
size_t xps_gru_seq_bwd_f32_workspace(int T, int B, int H, int ndir);
int xps_gru_seq_bwd_f32(const float* dy, const float* dhn, const float* y_ext, const float* saved,
                        const float* const* w_hh, const float* const* w_hh_t, float* dgi, float* dghn, float* dh0,
                        int T, int B, int H, int ndir, void* workspace, size_t workspace_bytes, void* stream);
/* Inter-layer dropout of a multi-layer GRU (torch.nn.GRU(dropout=p): nn_models/models.py:661-663, applied to the outputs of
 * every layer but the last) fused into the recurrence kernels of the register-resident shapes (H = 64 / 128;
 * xps_gru_seq_fused_dropout_supported says which): the forward kernel also writes y_drop (T x B x ndir*H) = y * keep / (1 - p)
 * with keep = the decisions xps_dropout_f32(seed) makes for the same flat index, the backward kernel takes dy as the gradient
 * w.r.t. y_drop and applies the same decisions while it loads it.  Same bits as the separate xps_dropout_f32 passes. */
int xps_gru_seq_fused_dropout_supported(int T, int B, int H, int ndir);
int xps_gru_seq_fwd_drop_f32(const float* gi, const float* const* w_hh, const float* const* b_hh,
                             const float* h0, float* y_ext, float* saved,
                             int T, int B, int H, int ndir, float* y_drop, float drop_p, uint64_t drop_seed,
                             void* workspace, size_t workspace_bytes, void* stream);
int xps_gru_seq_bwd_drop_f32(const float* dy, const float* dhn, const float* y_ext, const float* saved,
                             const float* const* w_hh, const float* const* w_hh_t, float* dgi, float* dghn, float* dh0,
                             int T, int B, int H, int ndir, float drop_p, uint64_t drop_seed,
                             void* workspace, size_t workspace_bytes, void* stream);
/* xps_gru_seq_fwd_f32 that ALSO writes XPS_FMT_SPLIT4 images (see xps_rowmap) of its outputs, from the epilogue that holds the
 * values (cluster-persistent shapes, 256 < H <= 512, bf16x3 mode: xps_gru_seq_fwd_images_supported): y_split (T + 2, B, ndir*H):
 * the image of y_ext, slot for slot -- h_prev of the dW_hh products; y_drop_split (T, B, ndir*H): the image of
 * dropout(y, drop_p, drop_seed) (drop_p = 0: of y) -- the input of the next layer's projection and dW_ih.  Either may be NULL.
 * Same bits as xps_split4_f32 over the finished tensors (one pass each over 168 MB at configs[3]'s shape, which this replaces).
 * Reference: the layer stack of nn_models/models.py:661-699 (torch.nn.GRU(dropout=p)). */
int xps_gru_seq_fwd_images_supported(int T, int B, int H, int ndir);
/* 1: for this shape the launch uses y_split as its in-kernel exchange buffer (H = 512, batch without pad trials): the image then
 * REPLACES the ring buffer's stores (one store per lane and round fewer) instead of adding one; callers ask for y_split by default
 * only where this holds. */
int xps_gru_seq_fwd_image_exchange_supported(int T, int B, int H, int ndir);
int xps_gru_seq_fwd_images_f32(const float* gi, const float* const* w_hh, const float* const* b_hh,
                               const float* h0, float* y_ext, float* saved, int T, int B, int H, int ndir,
                               float* y_split, float* y_drop_split, float drop_p, uint64_t drop_seed,
                               void* workspace, size_t workspace_bytes, void* stream);
/* The same backward with dgi / dghn written as XPS_FMT_SPLIT4 groups (see xps_rowmap): their only readers are GEMMs (weight
 * gradients, the layer's input gradient), and the BPTT kernels hold the bf16 hi / lo split of every gate gradient anyway.
 * bf16x3 mode; cluster-persistent shapes (256 < H <= 512) and the register-resident ones (H = 64 / 128):
 * xps_gru_seq_bwd_split4_supported.  drop_p > 0: the fused inter-layer dropout of xps_gru_seq_bwd_drop_f32 (resident shapes). */
int xps_gru_seq_bwd_split4_supported(int T, int B, int H, int ndir);
int xps_gru_seq_bwd_split4_f32(const float* dy, const float* dhn, const float* y_ext, const float* saved,
                               const float* const* w_hh, const float* const* w_hh_t, float* dgi, float* dghn, float* dh0,
                               int T, int B, int H, int ndir, float drop_p, uint64_t drop_seed,
                               void* workspace, size_t workspace_bytes, void* stream);

int xps_transpose_f32(const float* src, float* dst, int rows, int cols, void* stream);
/* 1..4 equally shaped matrices in one launch (host arrays of device pointers): both directions' W_hh^T */
int xps_transpose_batched_f32(const float* const* src, float* const* dst, int n, int rows, int cols, void* stream);

/* ------------------------------------------------------------------------- */
/* TemporalConv = Conv1d -> BatchNorm1d -> [ReLU] -> Dropout                    */
/* (nn_models/models.py:599-636).  The convolution itself is xps_gemm_nt_f32     */
/* over window rows; these are the fused normalisation passes.                   */
/*   y, out : [rows][F]  rows = T' * B                                           */
/*   stats  : [2F] sum and sum of squares over rows (all-reduced by the caller    */
/*            under data parallelism: SyncBN), count = global number of rows      */
/* ------------------------------------------------------------------------- */
/* num_batches_tracked: device int64 counter (BatchNorm1d buffer) incremented by one, or NULL */
int xps_bn_finalize_f32(const float* stats, double count, float* mean, float* rstd,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked,
                        float momentum, float eps, int F, void* stream);
int xps_bn_apply_f32(const float* y, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, const float* drop_mask, float drop_scale,
                     float* out, int64_t rows, int F, int relu, void* stream);
/* eval mode: running statistics */
/* xps_bn_finalize_f32 + xps_bn_apply_f32 in one launch (same values bit for bit; F <= 8192) */
int xps_bn_finalize_apply_f32(const float* y, const float* stats, double count, const float* gamma, const float* beta,
                              float* mean, float* rstd, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, float momentum, float eps, const float* drop_mask,
                              float drop_scale, float* out, int64_t rows, int F, int relu, void* stream);
int xps_bn_apply_eval_f32(const float* y, const float* running_mean, const float* running_var, float eps,
                          const float* gamma, const float* beta, float* out,
                          int64_t rows, int F, int relu, void* stream);
/* backward: pass 1 = g = dout * mask * relu'(out); sums[0:F] = sum g, sums[F:2F] = sum g*xhat
 *           (caller all-reduces sums under DP); dbeta_acc / dgamma_acc (optional, [F]): the LOCAL sums are
 *           also added into them (parameter gradients straight into their buffers); pass 2 = dy */
size_t xps_bn_bwd_workspace(int64_t rows, int F);
int xps_bn_bwd_reduce_f32(const float* dout, const float* out, const float* y, const float* mean,
                          const float* rstd, const float* drop_mask, float drop_scale, int relu,
                          float* sums, float* dbeta_acc, float* dgamma_acc, int64_t rows, int F,
                          void* workspace, size_t workspace_bytes, void* stream);
int xps_bn_bwd_apply_f32(const float* dout, const float* out, const float* y, const float* mean,
                         const float* rstd, const float* gamma, const float* drop_mask,
                         float drop_scale, int relu, const float* sums, double count,
                         float* dy, int64_t rows, int F, void* stream);

/* ------------------------------------------------------------------------- */
/* Decoder glue (nn_models/models.py:285-299,758-761)                           */
/* ------------------------------------------------------------------------- */
/* Fused autoregressive decoder: all L decode steps of a ONE-layer GRU decoder in one launch
 * (input-projection gather by token -> GRU cell, W_hh resident in registers -> Linear -> next token =
 * teacher token if flags[s] else first-max argmax).  Supported: H in {64, 128}, C <= 16, L <= 8
 * (xps_decoder_supported); other shapes use the composed entry points.
 *   table  [ntok][3H]  E W_ih^T + b_ih        teacher [B][L] (or NULL)   flags [L] DEVICE int32 (or NULL)
 *   logits [B][L][C]   tokens [L][B] input token of every step
 *   hs     [L+1][B][H] hidden state before step s (slot s) / after it (slot s+1)
 *   saved  [L][B][4H]  r, z, n, (W_hn h + b_hn)  (NULL in eval)
 * backward: dlogits [B][L][C] -> dgi [L][B][3H], dghn [L][B][H], dh0 [B][H]; weight gradients follow
 * from xps_gemm_tn_grouped_f32 / xps_scatter_rows_f32 over those buffers.                          */
int xps_decoder_supported(int H, int C, int L);
int xps_decoder_fwd_f32(const float* table, const float* w_hh, const float* b_hh, const float* h0,
                        const float* w_fc, const float* b_fc, const int64_t* teacher, const int32_t* flags,
                        float* logits, int64_t* tokens, float* hs, float* saved,
                        int B, int H, int C, int L, int ntok, int start_token, void* stream);
int xps_decoder_bwd_f32(const float* dlogits, const float* hs, const float* saved, const float* w_hh_t,
                        const float* w_fc, float* dgi, float* dghn, float* dh0,
                        int B, int H, int C, int L, void* stream);

/* Streaming (batch-1 .. 8 streams) inference for the realtime decoder (realtime_sim/realtime_nn_model.py
 * :153-170, one window per call): weight-streaming GEMV kernels, one wave per output row.
 *   x, h_prev, h_new, out are [S][.] with S = 1, 2, 4 or 8 rows ALLOCATED (S = B rounded up to a power
 *   of two); only the first B rows are written.  h_new must not alias h_prev.                       */
int xps_gemv_f32(const float* x, const float* W, const float* bias, float* out, int N, int K, int B,
                 void* stream);
int xps_gru_cell_gemv_f32(const float* x, int K, const float* w_ih, const float* w_hh, const float* b_ih,
                          const float* b_hh, const float* h_prev, float* h_new, int H, int B, void* stream);

/* out[b][:] = table[idx[b]][:]  (embedding / precomputed input projection rows) */
int xps_gather_rows_f32(const float* table, const int64_t* idx, float* out,
                        int B, int cols, int n_rows, void* stream);
/* dtable[r][:] (+)= sum_{b: idx[b]==r} dout[b][:]  (deterministic two-stage; n_rows <= 16) */
size_t xps_scatter_rows_f32_workspace(int B, int cols, int n_rows);
int xps_scatter_rows_f32(const float* dout, const int64_t* idx, float* dtable,
                         int B, int cols, int n_rows, int accumulate,
                         void* workspace, size_t workspace_bytes, void* stream);
/* next[b] = use_teacher[0] ? teacher[b*teacher_stride] : argmax_c logits[b][c]
 * (first maximal index, as torch.argmax); use_teacher is a DEVICE flag so the
 * decode loop has no host synchronisation.                                     */
int xps_next_token(const float* logits, int n_classes, const int64_t* teacher, int64_t teacher_stride,
                   const int32_t* use_teacher, int64_t* next, int B, void* stream);
/* Decode-step glue of a one-layer GRU decoder whose recurrence runs on the general GRU entry points (nn_models/models.py:
 * 285-301, 749-757): logits (B x C, C <= 16) = h W_fc^T + b_fc, next[b] = teacher token (device flag set) or the first
 * maximum of the logits, gi_next[b] = table[next[b]] (the token's row of the input-projection table, 3H floats) in one
 * launch.  next / gi_next may be NULL (last step: logits only). */
int xps_decoder_select_f32(const float* h, const float* w_fc, const float* b_fc, float* logits,
                           const int64_t* teacher, int64_t teacher_stride, const int32_t* use_teacher,
                           const float* table, int64_t* next, float* gi_next, int B, int H, int C, int ntok, void* stream);
/* Inverted dropout with a counter-based generator (no mask round trip through torch's RNG kernels):
 * mask[i] = (u_i >= p) in {0,1}, u_i a function of (seed, i) only;  if x != NULL also
 * out[i] = x[i] * mask[i] / (1 - p) in the same pass.  mask may be NULL when x is given: the backward pass then
 * regenerates the decisions by the same call on the incoming gradient (same seed), no mask tensor exists;
 * with a stored mask the backward is xps_mask_scale_f32.                                                 */
int xps_dropout_f32(const float* x, float* out, float* mask, int64_t n, float p, uint64_t seed, void* stream);
/* out = the XPS_FMT_SPLIT4 image (see xps_rowmap) of dropout(x) -- drop_p = 0: of x itself; else the values xps_dropout_f32(x,
 * seed) would write, decisions regenerated in the backward by xps_dropout_f32 on the gradient as before.  For tensors that only
 * GEMMs read (a layer input that exists only as the dropped output of the previous layer, weights once per forward pass):
 * the tile kernels then stage them without conversion arithmetic.  n % 4 == 0, 16-byte aligned buffers. */
int xps_split4_f32(const float* x, float* out, int64_t n, float drop_p, uint64_t seed, void* stream);
/* XPS_FMT_SPLIT4 image of a rows x cols fp32 matrix (leading dimension ldx) written with leading dimension ldo >= cols and ZERO
 * beyond column cols: a [k][n] GEMM operand whose rows are readable to a full 256-column tile (the skinny input-gradient product
 * of a layer with few input channels -- configs[3] layer 0: dx = dgi W_ih with In = 100 -- takes the LDS-DMA loop that way) */
int xps_split4_pad_f32(const float* x, int64_t ldx, int rows, int cols, float* out, int64_t ldo, void* stream);
/* out = x * mask * scale */
int xps_mask_scale_f32(const float* x, const float* mask, float scale, float* out, int64_t n, void* stream);
/* out = a + b (elementwise) */
int xps_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------- */
/* Loss and optimiser (nn_models/models.py:318-320 CrossEntropyLoss;            */
/* :373-389 AdamW; scripts/train_seq2seq.py:178 gradient_clip_val=0.5)           */
/* ------------------------------------------------------------------------- */
/* mean cross-entropy over `rows` rows of `n_classes` logits; row_loss [rows] scratch */
int xps_cross_entropy_fwd_f32(const float* logits, const int64_t* target, float* row_loss,
                              float* loss, int64_t rows, int n_classes, void* stream);
int xps_cross_entropy_bwd_f32(const float* logits, const int64_t* target, const float* gout,
                              float* dlogits, int64_t rows, int n_classes, void* stream);
/* Loss and UNIT gradient d(mean loss)/d(logits) in one launch (dlogits may be NULL: loss only).  `workspace`
 * (xps_cross_entropy_loss_grad_f32_workspace bytes, 8-byte aligned) holds a ticket word (its first word) and per-block partial sums; the ticket must
 * be ZERO at the first call and is left zero by every call: one zero-initialised buffer per stream serves all calls.            */
size_t xps_cross_entropy_loss_grad_f32_workspace(int64_t rows);
int xps_cross_entropy_loss_grad_f32(const float* logits, const int64_t* target, float* row_loss, float* loss,
                                    float* dlogits, void* workspace, size_t workspace_bytes, int64_t rows,
                                    int n_classes, void* stream);
/* CTC loss of the realtime CTC-RNN family (realtime_sim/realtime_nn_model.py:150, :213-224:
 * nn.CTCLoss(blank, reduction='mean', zero_infinity) on log_softmax(logits)).  logits [T][B][C] TIME-major raw
 * scores (the log-softmax is fused); targets [B][target_stride] int64 (padded); lengths int64 [B].
 * nll [B] per-sample negative log-likelihood; loss[0] = mean_b(nll_b / max(L_b, 1)); dlogits (optional, [T][B][C])
 * = d loss[0] / d logits.  One launch for the batch; workspace holds the alpha lattice.                          */
size_t xps_ctc_loss_f32_workspace(int T, int B, int max_target_len);
int xps_ctc_loss_f32(const float* logits, const int64_t* targets, int64_t target_stride,
                     const int64_t* input_lengths, const int64_t* target_lengths, int T, int B, int C,
                     int max_target_len, int blank, int zero_infinity, float* nll, float* loss,
                     float* dlogits, void* workspace, size_t workspace_bytes, void* stream);
/* sumsq[0] = sum g^2 over the flat gradient (deterministic two-stage) */
size_t xps_sumsq_f32_workspace(int64_t n);
int xps_sumsq_f32(const float* g, int64_t n, float* sumsq, void* workspace, size_t workspace_bytes,
                  void* stream);
/* AdamW over flat buffers with clip-by-global-norm folded in:
 *   coef = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6))  (max_norm <= 0: no clip)
 *   g *= coef (written back);  torch.optim.AdamW update with step count `step` (>= 1) */
int xps_adamw_f32(float* p, float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  void* stream);
/* the two above in TWO launches instead of three: gradient-norm partials, then clip + AdamW with the final fold of
 * the partials inside (sumsq[0], optional, receives the total = squared pre-clip norm); workspace as xps_sumsq_f32 */
int xps_clip_adamw_f32(float* p, float* g, float* m, float* v, int64_t n, float* sumsq, float max_norm, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int step, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Alignment (alignment/alignment_utils.py:42-61 cnd_avg; AlignCCA.py:235-285;   */
/* AlignMCCA.py:140-154 -> mvlearn MCCA; JointPCA.py:165-211; sklearn PCA at     */
/* nn_models/data_utils/datamodules.py:542-548)                                  */
/* ------------------------------------------------------------------------- */
/* Segmented condition mean.  data [N][row_len] (row_len = T*d), trials of
 * condition c are order[start[c] .. start[c+1]) (original trial order inside a
 * condition).  The sum runs trial after trial in the INPUT dtype, is divided by
 * the count in that dtype and stored as float64 -- np.mean semantics, bit for bit. */
int xps_cnd_avg_f32(const float* data, const int32_t* order, const int32_t* start, double* out,
                    int n_cond, int64_t row_len, void* stream);
int xps_cnd_avg_f64(const double* data, const int32_t* order, const int32_t* start, double* out,
                    int n_cond, int64_t row_len, void* stream);
/* column sums in float64 of an n x d matrix (float32 or float64 input) */
size_t xps_colsum_f64_workspace(int64_t n, int d);
int xps_colsum_f64(const void* X, int is_f32, int64_t ldx, int64_t n, int d, double* out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* Centred cross-covariance on the f64 MFMA:
 *   C[i][j] = sum_r (A[r][i] - mean_a[i]) * (B[r][j] - mean_b[j])        (da x db, float64)
 * A, B float32 or float64 (n x da, n x db); A == B gives the Gram / covariance.  */
size_t xps_xcov_f64_workspace(int64_t n, int da, int db);
int xps_xcov_f64(const void* A, int a_is_f32, int64_t lda, const double* mean_a,
                 const void* B, int b_is_f32, int64_t ldb, const double* mean_b,
                 double* C, int64_t ldc, int64_t n, int da, int db,
                 void* workspace, size_t workspace_bytes, void* stream);
/* One-sided Jacobi (Hestenes) on a column-major m x n float64 matrix W (n <= m is not
 * required): rotates column pairs of W and of V (n x n, column-major, identity on
 * entry) until all pairs are orthogonal.  On exit W = U * diag(sigma), V = right
 * singular vectors.  For a symmetric PSD matrix this is its eigendecomposition.
 * `sweeps` full sweeps are enqueued (no host sync); off[0] receives the largest
 * |cos angle| seen in the last sweep.                                          */
/* V may be NULL (rotations not accumulated: for a positive-definite matrix the eigenvectors are the normalised
 * columns of W).                                                                */
size_t xps_jacobi_f64_workspace(int n);
int xps_jacobi_sweeps_f64(double* W, int64_t ldw, double* V, int64_t ldv, int m, int n, int sweeps,
                          double* off, void* workspace, size_t workspace_bytes, void* stream);
/* Small problems (n <= 128 columns, n*m doubles within one workgroup's LDS): the WHOLE decomposition of each of
 * `batch` matrices (stride_w / stride_v doubles apart) in one launch, one workgroup per matrix; sweeps run
 * until the largest |cos angle| of a sweep is <= tol or max_sweeps.  V (optional) is SET to the accumulated
 * rotations (identity on entry is implied).  sweeps_done[batch], off[batch]: optional device outputs.       */
/* Whitening factors of a batch of small symmetric positive definite blocks: A_b = scale * R_b + shift * I = L L^T,
 * S_b = L^-T (upper triangular; S_b^T A_b S_b = I).  Replaces the per-view R_b^-1/2 the reference obtains inside
 * scipy.linalg.eigh(LHS, RHS) (mvlearn MCCA behind alignment/AlignMCCA.py:140-154; a generalised symmetric eigenproblem is
 * reduced with the Cholesky factor of RHS there too).  R_b at R + b * stride_r (row-major, ldr; the lower triangle is read),
 * A_b (optional, may be NULL) and S_b written at A + b * stride_a / S + b * stride_s.  n <= 136 (one workgroup, LDS resident).
 * info[b] = 0, or j + 1 when pivot j is not positive (S_b is then not written). */
int xps_chol_whiten_supported(int n);
int xps_chol_whiten_f64(const double* R, int64_t ldr, int64_t stride_r, double scale, double shift, double* A, int64_t lda,
                        int64_t stride_a, double* S, int64_t lds, int64_t stride_s, int n, int batch, int32_t* info,
                        void* stream);
int xps_jacobi_small_supported(int m, int n, int want_v);
int xps_jacobi_small_f64(double* W, int64_t ldw, int64_t stride_w, double* V, int64_t ldv, int64_t stride_v,
                         int m, int n, int batch, int max_sweeps, double tol, int32_t* sweeps_done, double* off,
                         void* stream);
/* Per-bin high-gamma features of the realtime pipeline (realtime_sim/realtime_processing.py:10-164 process_HG):
 * common average reference over the `good` channels (do_car) -> `bands` filters in scipy.signal.lfilter's direct-form-II-
 * transposed arithmetic (b, a: [bands][taps], normalised by a[0] inside; a == NULL: FIR) with the carried state
 * zi [bands][C][taps-1] updated in place (NULL: zero state) -> RMS over (time, bands) per channel in numpy's summation order.
 * data [C][Tn] float64.  Optional outputs: car_out [C][Tn], filtered [C][Tn][bands], power [C] (filtered and power are
 * exclusive: the power pass squares its copy in place; without `filtered` the copy lives in the workspace).            */
size_t xps_process_hg_f64_workspace(int C, int Tn, int bands);
int xps_process_hg_f64(const double* data, int C, int Tn, const uint8_t* good, const double* b, const double* a,
                       int bands, int taps, double* zi, int do_car, double* car_out, double* filtered,
                       double* power, void* workspace, size_t workspace_bytes, void* stream);
/* Y[r][:] = (X[r][:] - mean) @ Wt   X: n x d_in (float32 or float64), W: d_in x d_out float64,
 * Y float64 or float32.  Batched transform apply of every aligner.              */
int xps_apply_f64(const void* X, int x_is_f32, int64_t ldx, const double* mean, const double* W,
                  int64_t ldw, void* Y, int y_is_f32, int64_t ldy, int64_t n, int d_in, int d_out,
                  void* stream);
/* Batched C-SVC training on a precomputed kernel matrix (config-1 decode path: the one-vs-one linear SVMs sklearn's
 * SVC(kernel='linear') trains through libsvm; reference call sites decoders/cross_pt_decoders.py:29-38, scripts/aligned_decode_svm.py:262-263).
 * K: n_all x n_all kernel (Gram) matrix, float64, leading dimension ldk.  Problem p: the points idx[off[p] .. off[p+1]) of K, the
 * first npos[p] with label +1, the rest -1 (libsvm's binary sub-problem of a class pair).  One workgroup per problem runs libsvm's
 * SMO (WSS 2, eps stopping rule, calculate_rho) with alpha / gradient in LDS (max_points = the largest problem, at most
 * xps_svm_smo_f64_max_points()); cbound[off[p] + t] = the point's upper bound C * sample_weight (> 0: libsvm drops zero-weight
 * points before training, so does the caller).  Outputs: alpha[off[p] + t], rho[p], iterations[p].                          */
/* RBF kernel matrix from a Gram matrix (SVC(kernel='rbf'): scripts/aligned_decode_svm_ncv.py:313-317):
 * K[i][j] = exp(-gamma (na[i] + nb[j] - 2 G[i][j])), G = A B^T (m x n), na / nb the squared row norms of A / B -- libsvm's formula */
int xps_rbf_from_gram_f64(const double* G, int64_t ldg, const double* na, const double* nb, int m, int n, double gamma,
                          double* K, int64_t ldk, void* stream);
size_t xps_svm_smo_f64_max_points(void);
int xps_svm_smo_f64(const double* K, int64_t ldk, const int* idx, const int* off, const int* npos, int nprob, int max_points,
                    const double* cbound, double eps, int max_iter, double* alpha, double* rho, int* iterations, void* stream);
/* small dense float64 GEMM  C = op(A) op(B)  (row-major, op = transpose flag) */
int xps_dgemm_small(const double* A, int64_t lda, int ta, const double* B, int64_t ldb, int tb,
                    double* C, int64_t ldc, int M, int N, int K, void* stream);
/* the same product with the contraction split over workgroups (few output tiles, long K: the skinny products of the MCCA
 * eigensolve); deterministic partial slabs in `workspace` (xps_dgemm_splitk_workspace bytes), one reduce launch */
size_t xps_dgemm_splitk_workspace(int M, int N, int K);
int xps_dgemm_splitk(const double* A, int64_t lda, int ta, const double* B, int64_t ldb, int tb, double* C, int64_t ldc,
                     int M, int N, int K, void* workspace, size_t workspace_bytes, void* stream);
/* Chebyshev filter of an n x m block A (row-major, leading dimension m) by the symmetric n x n matrix C: the scaled three-term
 * recurrence of the top-k eigensolve behind AlignMCCA.fit (alignment/AlignMCCA.py:152-153 hands the generalised eigenproblem
 * to mvlearn / scipy.linalg.eigh): Y_1 = (C A - c A) sigma1 / e, Y_{j+1} = (C Y_j - c Y_j) 2 sigma_{j+1} / e - sigma_j sigma_{j+1} Y_{j-1},
 * sigma_{j+1} = 1 / (2 / sigma1 - sigma_j); `deg` products, all enqueued by this one call; out (n x m) receives Y_deg.       */
/* `steps` Lanczos steps (no reorthogonalisation) on the symmetric n x n matrix C from v0 / ||v0||: alpha[steps], beta[steps]
 * (device arrays) = diagonal / off-diagonal of the tridiagonal matrix whose extreme Ritz values bound the spectrum for the filter */
size_t xps_lanczos_f64_workspace(int n);
int xps_lanczos_f64(const double* C, int64_t ldc, int n, int steps, const double* v0, double* alpha, double* beta,
                    void* workspace, size_t workspace_bytes, void* stream);
size_t xps_cheb_filter_f64_workspace(int n, int m);
int xps_cheb_filter_f64(const double* C, int64_t ldc, int n, const double* A, int m, int deg, double c, double e, double sigma1,
                        double* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* Training-set augmentations on (trial x time x channel) fp32 tensors           */
/* (nn_models/data_utils/augmentations.py:13-90).  The caller makes the random    */
/* draw exactly as the reference does (numpy / torch global generators) and       */
/* passes it in; x and out are [N][T][C], out != x for shift / warp.               */
/*   time_shift : out = torch.roll(x, shift, dims=1)                 (:51-62)      */
/*   time_mask  : out = x with [start, start + size) along time zeroed (:32-48)     */
/*   scale      : out = x * scale                                      (:79-89)    */
/*   jitter     : out = x + noise * level  (noise: the N(0,1) draw)    (:65-76)    */
/*   time_warp  : scipy.ndimage.zoom(order=1) to T2 samples, then torchvision       */
/*                Resize back to T (bilinear, antialias), fused          (:13-29)   */
/* ------------------------------------------------------------------------- */
int xps_aug_time_shift_f32(const float* x, float* out, int N, int T, int C, int shift, void* stream);
int xps_aug_time_mask_f32(const float* x, float* out, int N, int T, int C, int start, int size, void* stream);
int xps_aug_scale_f32(const float* x, float* out, int64_t n, float scale, void* stream);
int xps_aug_jitter_f32(const float* x, const float* noise, float* out, int64_t n, float level, void* stream);
int xps_aug_time_warp_f32(const float* x, float* out, int N, int T, int C, int T2, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XPS_H */
