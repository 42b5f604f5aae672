// Dense fp32 GEMMs in two product precisions (xps_set_gemm_precision): the f32-input MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chains, 64 FLOP/clk/SIMD = 157 TFLOP/s) described
// below, or the bf16 split-product pipeline of xps_gemm_tile.h (template flag BF: operands split
// hi/lo on the way into LDS, three v_mfma_f32_32x32x16_bf16 per product; the default).
//
// One LDS-tiled kernel, 128 x 128 x 16 block tile, 4 waves each owning a 64 x 64
// sub-tile (2 x 2 MFMA 32x32 tiles).  Both operands are staged k-major in LDS
// ([k][m] / [k][n], row stride 132 floats) so that the MFMA operand reads are
// conflict-free ds_read_b32 (lanes 0-31 consecutive floats; lanes 32-63 the next k
// row).  Global loads are 16-byte vectors along whichever index is contiguous in
// memory; the next tile is fetched into registers while the current one is
// multiplied (one barrier per k-tile, two LDS buffers).
//
// Operand forms (template flags):  AK = A is contiguous along k ([m][k]),
// otherwise [k][m];  BK likewise for B.  NT = <1,1>, NN = <1,0>, TN = <0,0>.
// Rows are addressed through a two-level row map (see xps.h) so the same kernel
// reads convolution windows of a (trial, time, channel) tensor and writes
// time-major outputs.
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include "xps_common.h"
#include "xps_gemm_tile.h"
#include "xps_gemm_big.h"
#include "xps_gemm_dma.h"
using namespace xps_tile;
#ifndef XPS_GEMM_DEFAULT_MODE
#define XPS_GEMM_DEFAULT_MODE 1
#endif
namespace {

// C (+)= A B (+ A2 B2) + bias.  MI = 1: 64 x 128 tiles (twice the blocks, for grids that would not fill
// the chip with 128 x 128 tiles);  A2/B2: an optional second operand pair with the same row maps (the two
// directions of a bidirectional layer summed in registers instead of a second accumulate pass).
// 64-row tiles of the bf16 pipeline use 36 KB of LDS per block, so four blocks would fit a CU if the kernel kept to 128
// registers (-DXPS_BF_MI1_WAVES=4).  Measured: the projections gain 3-6 %, the two-operand input-gradient form loses 15 %
// to spills, the step is unchanged — three blocks per CU stay the default.
#ifndef XPS_BF_MI1_WAVES
#define XPS_BF_MI1_WAVES XPS_GEMM_WAVES
#endif
template <bool AK, bool BK, int MI, bool EDGE = false, bool BF = false, bool PRE = false>      // PRE: XPS_FMT_SPLIT4 operands (bf16x3 only)
__global__ __launch_bounds__(256, (BF && MI == 1) ? XPS_BF_MI1_WAVES : XPS_GEMM_WAVES) void gemm_f32_kernel(
    const float* __restrict__ A, RowMap ra, const float* __restrict__ B, RowMap rb,
    const float* __restrict__ A2, const float* __restrict__ B2, int K2,
    float* __restrict__ C, RowMap rc, const float* __restrict__ bias,
    int M, int N, int K, int kchunk, long long slab_stride, int accumulate, int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) TileMem<BF, MI> mem;
    // 1-D grid, logical order (k-split, m-tile, n-tile): the n-tiles of an A panel run on one XCD
    const int tiles_n = (N + BN - 1) / BN, tiles = tiles_n * ((M + 64 * MI - 1) / (64 * MI));
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int z = lid / tiles, t = lid % tiles;
    const int kbeg = z * kchunk;
    const int kend = min(K, kbeg + kchunk);
    const int m0 = (t / tiles_n) * (64 * MI), n0 = (t % tiles_n) * BN;
    f32x16 acc[MI][2];
    zero_acc<MI>(acc);
    float nocs = 0.f;
    f32x4 nocs4 = {0.f, 0.f, 0.f, 0.f};
    gemm_accumulate_any<AK, BK, MI, EDGE, BF, PRE>(acc, nocs, nocs4, false, A, ra, B, rb, M, N, K, m0, n0, kbeg, kend, vecA, vecB, mem);
    if (A2) gemm_accumulate_any<AK, BK, MI, EDGE, BF, PRE>(acc, nocs, nocs4, false, A2, ra, B2, rb, M, N, K2, m0, n0, 0, K2, vecA, vecB, mem);
    gemm_store<MI>(acc, C + (long long)z * slab_stride, rc, bias, M, N, m0, n0, accumulate);
}

// Few-row products (M <= 16: the decoder's token table E W_ih^T and its input gradient).  An MFMA tile
// would be > 87 % padding and, with one or two blocks, pure load latency.  Here a 1024-thread block owns 64
// columns x 16 k-slices (one wave each).  A (at most 16 x 1024 per chunk) is staged in LDS with coalesced
// loads and read back as wave-uniform 16-byte broadcasts; every lane keeps all M accumulators; the sixteen
// k-slices are folded through LDS in a fixed order.
constexpr int SM_MAXM = 16, SM_SLICES = 16, SM_KC = 1024;
template <bool AK, bool BK>
__global__ __launch_bounds__(1024) void gemm_small_kernel(
    const float* __restrict__ A, RowMap ra, const float* __restrict__ B, RowMap rb,
    const float* __restrict__ A2, const float* __restrict__ B2, int K2,
    float* __restrict__ C, RowMap rc, const float* __restrict__ bias, int M, int N, int K, int accumulate, int vecB) {
    __shared__ __attribute__((aligned(16))) float As[SM_MAXM][SM_KC];
    __shared__ float red[SM_SLICES][SM_MAXM][64];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int n = blockIdx.x * 64 + l;
    float acc[SM_MAXM];
#pragma unroll
    for (int m = 0; m < SM_MAXM; ++m) acc[m] = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        const float* __restrict__ Ap = pass ? A2 : A;
        const float* __restrict__ Bp = pass ? B2 : B;
        const int Kp = pass ? K2 : K;
        if (!Ap) break;
        const bool plain_b = rb.rpg >= Kp;                       // k rows of B addressed by a plain leading dimension
        for (int kc = 0; kc < Kp; kc += SM_KC) {
            const int kn = min(SM_KC, Kp - kc);
            __syncthreads();
            for (int e = tid; e < SM_MAXM * kn; e += 1024) {     // stage A[0..M) x [kc, kc+kn); rows >= M are zero
                const int m = e / kn, k = e - m * kn;
                As[m][k] = m < M ? (AK ? Ap[ra.off(m) + kc + k] : Ap[ra.off(kc + k) + m]) : 0.f;
            }
            __syncthreads();
            const int kq = (((kn + SM_SLICES - 1) / SM_SLICES) + 3) & ~3;       // k-slice length, multiple of 4
            const int k0 = w * kq, k1 = min(kn, k0 + kq);
            if (n < N) {
                const long long boff = BK ? rb.off(n) + kc : (long long)n;
                int k = k0;
                for (; k + 3 < k1; k += 4) {
                    float b[4];
                    if (BK && vecB) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(Bp + boff + k);
                        b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            b[j] = BK ? Bp[boff + k + j]
                                      : Bp[(plain_b ? (long long)(kc + k + j) * rb.ld : rb.off(kc + k + j)) + boff];
                    }
#pragma unroll
                    for (int m = 0; m < SM_MAXM; ++m) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(&As[m][k]);        // same address in every lane
                        acc[m] += a.x * b[0]; acc[m] += a.y * b[1]; acc[m] += a.z * b[2]; acc[m] += a.w * b[3];
                    }
                }
                for (; k < k1; ++k) {
                    const float b = BK ? Bp[boff + k] : Bp[(plain_b ? (long long)(kc + k) * rb.ld : rb.off(kc + k)) + boff];
#pragma unroll
                    for (int m = 0; m < SM_MAXM; ++m) acc[m] += As[m][k] * b;
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < SM_MAXM; ++m) red[w][m][l] = acc[m];
    __syncthreads();
    if (w < M && n < N) {                        // wave w folds output row w
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < SM_SLICES; ++s) v += red[s][w][l];
        if (bias) v += bias[n];
        float* c = C + rc.off(w) + n;
        if (accumulate) v += *c;
        *c = v;
    }
}

// Same A, up to 4 different (B, bias, C): the input projections of all directions of a layer in ONE launch
// (blockIdx.z selects the problem) — twice the blocks per launch halves the under-filled tail round.
struct NtMulti {
    const float* B[4];
    const float* bias[4];
    float* C[4];
};
template <int MI, bool EDGE = false, bool BF = false, bool PRE = false>
__global__ __launch_bounds__(256, (BF && MI == 1) ? XPS_BF_MI1_WAVES : XPS_GEMM_WAVES) void gemm_nt_multi_kernel(
    const float* __restrict__ A, RowMap ra, NtMulti pm, RowMap rb, RowMap rc, int M, int N, int K, int nprob,
    int vecA, int vecB) {
    __shared__ __attribute__((aligned(16))) TileMem<BF, MI> mem;
    // logical order (m-tile, problem, n-tile): all blocks that read one A panel are neighbours on one XCD
    const int tiles_n = (N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_m = nprob * tiles_n, rem = lid % per_m;
    const int z = rem / tiles_n;
    const int m0 = (lid / per_m) * (64 * MI), n0 = (rem % tiles_n) * BN;
    f32x16 acc[MI][2];
    zero_acc<MI>(acc);
    float nocs = 0.f;
    f32x4 nocs4 = {0.f, 0.f, 0.f, 0.f};
    gemm_accumulate_any<true, true, MI, EDGE, BF, PRE>(acc, nocs, nocs4, false, A, ra, pm.B[z], rb, M, N, K, m0, n0, 0, K, vecA, vecB, mem);
    gemm_store<MI>(acc, pm.C[z], rc, pm.bias[z], M, N, m0, n0, 0);
}


// ---------------------------------------------------------------------------------------------------------------
// Weight-stationary input projection (bf16x3 mode):  C_z = A B_z^T + bias_z  for the nprob weight matrices of a GRU layer,
// K <= 256.  The tile kernels above re-stage (and re-split) the 128 x K weight tile in every one of the M / 64 row tiles and
// spend as long outside their 16-iteration k loop as inside it (2.0-2.2 TB/s on the cfg-2 projections).  Here a
// persistent 8-wave block keeps a 256-column slice of the stacked weights [problem][n] in REGISTERS for the whole launch
// (wave w: 32 columns, all of K as hi / lo bf16 fragments: 8 KT registers, KT = 16-wide k-steps) and streams its share of the
// 64-row tiles of A through a double-buffered LDS image (rows of KP = 16 KT bf16 hi + lo, row stride = 8 dwords mod 64); the
// next tile's global loads are in flight while the current one is multiplied.  Per accumulator the products are issued in
// the order of the tile kernels (k-step by k-step: lo*hi, hi*lo, hi*hi), so the results have the SAME BITS.
// Blocks: logical id (xcd_remap) -> (row group, slice): the slices of a row group are neighbours on one XCD and read the same
// A tiles from that L2.  A may be an XPS_FMT_SPLIT4 operand, so may the weights.
// Measured (tools/bench_proj.py, cfg-2 shapes, 2 directions): 40960 x 384 x 256: 90 -> 73-83 us; K = 100: 48 -> 42 us;
// 40960 x 192 x 128: 41 -> 27 us.  Builds without the C stores / without the MFMAs run the first shape in 58 / 42-50 us: the
// phases of a tile (stage, multiply, store) do not overlap -- all eight waves move in step through one barrier per tile; the
// software pipeline that was tried next -- 32-row tiles (to make room for a pending tile beside 128 weight registers), the next
// tile's split + LDS stores and the previous tile's C stores spread over the k-steps of the current one, one barrier per tile;
// same bits, 244 registers -- ran SLOWER (82-93 us): one accumulator chain per wave instead of two, twice the barriers per
// byte.  Left for the next round: two independent wave teams per CU (anti-phase), or the pipeline with 64-row tiles for K <= 192.
// Round 4, the C stores (the kernel writes 503 MB of gi for configs[3]'s layer 0 at 3.4 TB/s through 32 scalar stores per lane and
// tile, each wave-instruction two FULL 128-byte lines): 16-byte stores from SWAPPED MFMA operands (a lane then holds four
// consecutive columns of one row; 8 store instructions instead of 32; same bits) were built and measured -- every instruction
// then touches 32 rows x 32 bytes, and the launches take 125-130 / 112-117 / 56 us instead of 85-92 / 43-45 / 28 us on the three
// shapes above (configs[3] step + 0.3 ms): full lines per instruction matter more than the instruction count.  Not kept.
template <int KT>
__global__ __launch_bounds__(512, 1) void proj_ws_kernel(const float* __restrict__ A, long long lda, NtMulti pm, long long ldb, long long ldc,
                                                         int M, int N, int K, int nprob, int preA, int preB) {
    constexpr int KP = 16 * KT;                       // padded k extent (multiple of 32)
#ifndef XPS_WS_PAD
#define XPS_WS_PAD 4
#endif
    // LDS row stride in dwords (2 bf16 each), = 4 (mod 64): the b128 fragment reads (32 rows x one k-half per half-wave; lane
    // groups {0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS) then touch 16 distinct 4-bank groups per cycle
    constexpr int LDH = ((KP / 2 + 63) / 64) * 64 + XPS_WS_PAD;
    constexpr int NV = KP / 32;                       // 16-byte vectors per thread and tile (64 rows x KP / 4 vectors / 512 threads)
    extern __shared__ __attribute__((aligned(16))) unsigned char ws_smem[];
    unsigned* lds = reinterpret_cast<unsigned*>(ws_smem);          // [buffer 2][plane hi, lo][64 rows][LDH dwords]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NC = nprob * N, S = (NC + 255) / 256;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int slice = lid % S, rg = lid / S;
    const int nb = (gridDim.x - slice + S - 1) / S;   // blocks that serve this slice = stride of the row-tile loop
    const int c0 = slice * 256 + wave * 32;           // this wave's first stacked column
    const bool active = c0 < NC;
    const int z = active ? c0 / N : 0, n0 = active ? c0 % N : 0;
    const int li = lane & 31, lk = lane >> 5;

    // resident weight fragments: B operand of v_mfma_f32_32x32x16_bf16 for k-step kk = column n0 + li, k = 16 kk + 8 lk + 0..7
    bf16x8 wh[KT], wl[KT];
    {
        const float* wrow = pm.B[z] + (long long)(n0 + li) * ldb;
#pragma unroll
        for (int kk = 0; kk < KT; ++kk) {
            const int k = 16 * kk + 8 * lk;
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (active && k < K) v0 = *reinterpret_cast<const f32x4*>(wrow + k);
            if (active && k + 4 < K) v1 = *reinterpret_cast<const f32x4*>(wrow + k + 4);
            bf16x4 h0, l0, h1, l1;
            stage_split(v0, preB != 0, h0, l0);
            stage_split(v1, preB != 0, h1, l1);
            wh[kk] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            wl[kk] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
    float bv = 0.f;
    if (active && pm.bias[z]) bv = pm.bias[z][n0 + li];

    const int tiles = (M + 63) / 64;
    // thread -> vectors v = tid + 512 j of a tile: row = v / (KP / 4), k = 4 (v % (KP / 4))
    auto load_tile = [&](int t, f32x4 (&r)[NV]) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = tid + 512 * j, row = t * 64 + v / (KP / 4), k = 4 * (v % (KP / 4));
            r[j] = (row < M && k < K) ? *reinterpret_cast<const f32x4*>(A + (long long)row * lda + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tile = [&](const f32x4 (&r)[NV], int buf) {
        unsigned* hi = lds + (size_t)buf * 2 * 64 * LDH;
        unsigned* lo = hi + 64 * LDH;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int v = tid + 512 * j, row = v / (KP / 4), kd = 2 * (v % (KP / 4));       // dword offset of 4 bf16
            bf16x4 h, l;
            stage_split(r[j], preA != 0, h, l);
            *reinterpret_cast<bf16x4*>(hi + row * LDH + kd) = h;
            *reinterpret_cast<bf16x4*>(lo + row * LDH + kd) = l;
        }
    };

    f32x4 raw[NV];
    int t = rg;
    if (t < tiles) load_tile(t, raw);
    for (int it = 0; t < tiles; t += nb, ++it) {
        const int buf = it & 1;
        store_tile(raw, buf);
        if (t + nb < tiles) load_tile(t + nb, raw);
        __syncthreads();              // tile `it` is in LDS; everyone has finished reading buffer `buf` two iterations ago
        if (active) {
            const unsigned* hi = lds + (size_t)buf * 2 * 64 * LDH;
            const unsigned* lo = hi + 64 * LDH;
            f32x16 acc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int off = (32 * i + li) * LDH + 8 * kk + 4 * lk;
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(hi + off);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(lo + off);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh[kk], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl[kk], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh[kk], acc[i], 0, 0, 0);
                }
            }
            float* cbase = pm.C[z] + n0 + li;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = t * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (row < M) __builtin_nontemporal_store(acc[i][r] + bv, cbase + (long long)row * ldc);
                }
        }
    }
}
constexpr int proj_ws_lds(int KT) { return 2 * 2 * 64 * (((16 * KT / 2 + 63) / 64) * 64 + XPS_WS_PAD) * 4; }

// Grouped TN GEMM: up to TN_MAXP weight-gradient problems  C_p = A_p^T B_p  (+ column sums of A_p
// from the LDS copy of the A tiles = the bias gradient) in ONE launch; every block owns one
// (problem, tile, k-split) and writes a partial slab.
constexpr int TN_MAXP = 12;
struct TnProb {
    const float* A; const float* B; float* C; float* colsum;
    RowMap ra, rb, rc;
    int M, N, K, accumulate;
    int splits, kchunk, tiles_n, vecA, vecB;
    long long slab_off;       // float offset of this problem's slabs in the workspace
    int block_start;          // first flat block id (of the launch that serves this problem)
    int big;                  // != 0: served by the 256 x 256 tile kernel (gemm_big_tn_kernel), same slab layout; 2: both operands
                              // are XPS_FMT_SPLIT4 and the block runs the LDS-DMA k loop (xps_gemm_dma.h)
};
struct TnGroup {
    TnProb p[TN_MAXP];
    int n;
    int total_blocks;         // blocks of the 128 x 128 launch (problems with big == 0)
    int total_blocks_big;     // blocks of the 256 x 256 launch (problems with big != 0)
    long long total_out;      // reduce outputs: sum of tiles * TN_TILE (padded tiles) + M column sums
    // uniform != 0: every problem has the same K and the same k-split count; blocks are then ordered (k-split, problem, tile)
    // with tile_start[p] = tiles of the problems before p and tiles_all = tiles of all problems
    int uniform, tiles_all;
    int tile_start[TN_MAXP];
};

constexpr int TN_TILE = BM * BN;       // floats of one output tile in a slab
// floats of one split's slab: the tiles, then the column sums (padded so that every slab stays 16-byte aligned)
__host__ __device__ inline long long tn_split_stride(long long ntiles, int M, bool has_colsum) {
    return ntiles * TN_TILE + (has_colsum ? ((M + 3) & ~3) : 0);
}
// slab tile in register order: [wave 4][i 2][j 2][q 4][lane 64][c 4], accumulator register = 4 q + c
__device__ inline void slab_store(const f32x16 (&acc)[2][2], float* __restrict__ tile) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* base = tile + wave * 4096 + lane * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(base + ((i * 2 + j) * 4 + q) * 256));
            }
}
// (row, col) inside the 128 x 128 tile of slab element e (inverse of slab_store + the MFMA C layout)
__device__ inline void slab_decode(int e, int& row, int& col) {
    const int wave = e >> 12, ij = (e >> 10) & 3, q = (e >> 8) & 3, lane = (e >> 2) & 63, c = e & 3;
    const int reg = 4 * q + c, li = lane & 31, lk = lane >> 5;
    row = (wave >> 1) * 64 + (ij >> 1) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lk;
    col = (wave & 1) * 64 + (ij & 1) * 32 + li;
}

template <bool EDGE, bool BF = false, bool PRE = false>
__global__ __launch_bounds__(256, XPS_GEMM_WAVES) void gemm_tn_grouped_kernel(TnGroup g, float* __restrict__ ws) {
    __shared__ __attribute__((aligned(16))) TileMem<BF> mem;
    // Logical order, uniform groups (the weight gradients of a GRU layer: same K = T' x B rows for every problem):
    // (k-split, problem, tile) -- ALL blocks that read rows [kbeg, kend) of the operands are neighbours on one XCD and run at
    // the same time, so a panel shared by several tiles AND several problems (dgi feeds dW_ih and dW_hh, x feeds both
    // directions, h_prev the r/z and the n rows) comes from HBM once and from that XCD's L2 afterwards.
    // Otherwise (problem, k-split, tile): only the tiles of one problem's k-chunk are neighbours.
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0, local;
    if (g.uniform) {
        const int zz = lid / g.tiles_all, rem = lid - zz * g.tiles_all;
#pragma unroll 1
        for (int i = 1; i < g.n; ++i)
            if (rem >= g.tile_start[i]) pi = i;
        const int nt = ((g.p[pi].M + BM - 1) / BM) * g.p[pi].tiles_n;
        local = zz * nt + (rem - g.tile_start[pi]);
    } else {
#pragma unroll 1
        for (int i = 0; i < g.n; ++i)
            if (!g.p[i].big && lid >= g.p[i].block_start) pi = i;
        local = lid - g.p[pi].block_start;
    }
    const TnProb& P = g.p[pi];
    const int tiles_m = (P.M + BM - 1) / BM;
    const int ntiles = tiles_m * P.tiles_n;
    const int tile = local % ntiles, z = local / ntiles;
    const int tm = tile / P.tiles_n, tn = tile % P.tiles_n;
    const int kbeg = z * P.kchunk;
    const int kend = min(P.K, kbeg + P.kchunk);
    // One split's slab: [tile][TN_TILE floats in REGISTER order] then [M column sums].  The slab is private scratch, so a
    // tile is written exactly as the accumulators sit in the lanes (16-byte stores, 4 per 32x32 MFMA tile instead of 16
    // 4-byte ones: the store phase of a block was ~24 % of its time); the reduce pass undoes the permutation.
    const long long split_stride = tn_split_stride(ntiles, P.M, P.colsum != nullptr);
    float* slab = ws + P.slab_off + (long long)z * split_stride;
    float* cs = (P.colsum && tn == 0) ? slab + (long long)ntiles * TN_TILE + tm * BM : nullptr;
    f32x16 acc[2][2];
    zero_acc<2>(acc);
    float csum = 0.f;
    f32x4 csum4 = {0.f, 0.f, 0.f, 0.f};
    gemm_accumulate_any<false, false, 2, EDGE, BF, PRE>(acc, csum, csum4, cs != nullptr, P.A, P.ra, P.B, P.rb, P.M, P.N, P.K, tm * BM, tn * BN,
                                                   kbeg, kend, P.vecA, P.vecB, mem);
    slab_store(acc, slab + (long long)tile * TN_TILE);
    if (BF && cs) {
        // bf16 pipeline: thread (k row = tid / 32, x group = tid % 32) holds the sums of columns 4 xg .. 4 xg + 3 over its
        // k rows; lanes l and l + 32 of a wave share the x group, the four waves are folded through LDS in a fixed order
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        float* red = reinterpret_cast<float*>(&mem);
#pragma unroll
        for (int j = 0; j < 4; ++j) csum4[j] += __shfl_xor(csum4[j], 32);
        __syncthreads();
        if (lane < 32) *reinterpret_cast<f32x4*>(&red[wave * 128 + lane * 4]) = csum4;
        __syncthreads();
        if (tid < 128 && tm * BM + tid < P.M) cs[tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
    } else if (cs) {                            // fold the two k-groups of the column sums through LDS
        const int tid = threadIdx.x;
        float* red = reinterpret_cast<float*>(&mem);
        __syncthreads();
        red[tid] = csum;
        __syncthreads();
        if (tid < 128 && tm * BM + tid < P.M) cs[tid] = red[tid] + red[tid + 128];
    }
}

// Slab reduction.  One block = 64 consecutive outputs x 4 split lanes (one wave each): wave w sums slabs
// w, w+4, ... with 8 loads in flight, then the four partial sums are folded in a fixed order through LDS
// (deterministic: no atomics, the order depends only on the split count).
constexpr int RED_OUT = 64;
__device__ inline float slab_sum(const float* __restrict__ p, long long stride, int splits, int w) {
    float s = 0.f;
    int z = w;
    for (; z + 28 < splits; z += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(long long)(z + 4 * u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < splits; z += 4) s += p[(long long)z * stride];
    return s;
}

// 16-byte version of slab_sum: four consecutive slab elements per lane
__device__ inline f32x4 slab_sum4(const float* __restrict__ p, long long stride, int splits, int w) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int z = w;
    for (; z + 28 < splits; z += 32) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (long long)(z + 4 * u) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < splits; z += 4) s += *reinterpret_cast<const f32x4*>(p + (long long)z * stride);
    return s;
}

// One block = 64 lanes x 4 consecutive slab elements x 4 split lanes (one wave each); every size in the slab layout is a
// multiple of 4 floats, so a lane's four elements belong to one tile (same q, lane: accumulator registers 4q .. 4q + 3) or
// to the column-sum tail of one problem.
__global__ __launch_bounds__(256) void gemm_tn_grouped_reduce(TnGroup g, const float* __restrict__ ws) {
    __shared__ f32x4 part[4][RED_OUT];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long idx = ((long long)blockIdx.x * RED_OUT + l) * 4;
    const bool live = idx < g.total_out;
    int pi = 0;
    long long base = 0;
    if (live) {
#pragma unroll 1
        for (int i = 0; i < g.n; ++i) {
            const long long sz = tn_split_stride((long long)((g.p[i].M + BM - 1) / BM) * g.p[i].tiles_n, g.p[i].M, g.p[i].colsum != nullptr);
            if (idx < base + sz) { pi = i; break; }
            base += sz;
        }
    }
    const TnProb& P = g.p[pi];
    const long long tile_floats = (long long)((P.M + BM - 1) / BM) * P.tiles_n * TN_TILE;
    const long long split_stride = tn_split_stride(tile_floats / TN_TILE, P.M, P.colsum != nullptr);
    const long long e = idx - base;                       // [0, tile_floats): products in slab order, then M column sums
    part[w][l] = live ? slab_sum4(ws + P.slab_off + e, split_stride, P.splits, w) : f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (w == 0 && live) {
        const f32x4 s4 = (part[0][l] + part[1][l]) + (part[2][l] + part[3][l]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float* dst = nullptr;
            bool is_cs = false;
            if (e < tile_floats) {
                const int tile = (int)(e / TN_TILE);
                int r, cc;
                slab_decode((int)(e % TN_TILE) + c, r, cc);
                const int row = (tile / P.tiles_n) * BM + r, col = (tile % P.tiles_n) * BN + cc;
                if (row < P.M && col < P.N)                                        // else: padding of an edge tile
                    dst = P.C + (P.rc.rpg >= P.M ? (long long)row * P.rc.ld : P.rc.off(row)) + col;
            } else {
                is_cs = true;
                if (e - tile_floats + c < P.M) dst = P.colsum + (e - tile_floats + c);   // else: alignment padding
            }
            if (dst) {
                float s = s4[c];
                // accumulate bit 0: products AND column sums add to their destinations; bit 1: the column sums only
                if ((P.accumulate & 1) || ((P.accumulate & 2) && is_cs)) s += *dst;
                *dst = s;
            }
        }
    }
}

// The same reduction with ONE thread per 16-byte vector summing every split in index order (eight loads in flight): for groups
// with few splits (the 256-tile launches of configs[3]: 3-10 slabs) the four split lanes + LDS fold of the kernel above leave
// one or two loads in flight per thread (92 us for 147 MB); XPS_TN_REDUCE=lanes|flat overrides the choice (flat: <= 32 splits;
// measured on the whole step with XPS_TN_BLOCKS = 768 / 512 / 384 / 256 x lanes / flat: 512 + flat wins on both workloads).
__global__ __launch_bounds__(256) void gemm_tn_grouped_reduce_flat(TnGroup g, const float* __restrict__ ws) {
    const long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    const bool live = idx < g.total_out;
    int pi = 0;
    long long base = 0;
    if (live) {
#pragma unroll 1
        for (int i = 0; i < g.n; ++i) {
            const long long sz = tn_split_stride((long long)((g.p[i].M + BM - 1) / BM) * g.p[i].tiles_n, g.p[i].M, g.p[i].colsum != nullptr);
            if (idx < base + sz) { pi = i; break; }
            base += sz;
        }
    }
    const TnProb& P = g.p[pi];
    const long long tile_floats = (long long)((P.M + BM - 1) / BM) * P.tiles_n * TN_TILE;
    const long long split_stride = tn_split_stride(tile_floats / TN_TILE, P.M, P.colsum != nullptr);
    const long long e = idx - base;                       // [0, tile_floats): products in slab order, then M column sums
    if (live) {
        // every split in index order by ONE thread, eight 16-byte loads in flight (few splits: no lanes to fold, no LDS)
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
        const float* src = ws + P.slab_off + e;
        int z = 0;
        for (; z + 8 <= P.splits; z += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (long long)(z + u) * split_stride);
#pragma unroll
            for (int u = 0; u < 8; ++u) s4 += v[u];
        }
        {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = z + u < P.splits ? *reinterpret_cast<const f32x4*>(src + (long long)(z + u) * split_stride) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) s4 += v[u];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float* dst = nullptr;
            bool is_cs = false;
            if (e < tile_floats) {
                const int tile = (int)(e / TN_TILE);
                int r, cc;
                slab_decode((int)(e % TN_TILE) + c, r, cc);
                const int row = (tile / P.tiles_n) * BM + r, col = (tile % P.tiles_n) * BN + cc;
                if (row < P.M && col < P.N)                                        // else: padding of an edge tile
                    dst = P.C + (P.rc.rpg >= P.M ? (long long)row * P.rc.ld : P.rc.off(row)) + col;
            } else {
                is_cs = true;
                if (e - tile_floats + c < P.M) dst = P.colsum + (e - tile_floats + c);   // else: alignment padding
            }
            if (dst) {
                float s = s4[c];
                // accumulate bit 0: products AND column sums add to their destinations; bit 1: the column sums only
                if ((P.accumulate & 1) || ((P.accumulate & 2) && is_cs)) s += *dst;
                *dst = s;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 256 x 256 tiles on 8 waves (xps_gemm_big.h): large interior shapes in bf16 split-product mode.
// ---------------------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) unsigned char big_smem[];

// C (+)= A B (+ A2 B2) + bias, plain leading dimensions.  1-D grid, logical order (m-tile, n-tile).
// DEEP: 32-deep LDS stages (k ranges multiples of 32), see xps_gemm_big.h
// FMT >= 0 (16-deep stages): the operand formats are compile-time (bit 0: A, bit 1: B are XPS_FMT_SPLIT4) -- ONE k loop in the
// kernel; FMT < 0: the run-time flags select among four loops (kernels that have the registers for it)
template <bool AK, bool BK, bool DEEP, int FMT = -1>
__device__ inline void big_accumulate(f32x16 (&acc)[4][2], f32x4& csum, bool want_csum, const float* __restrict__ A, long long lda,
                                      const float* __restrict__ B, long long ldb, int m0, int n0, int kbeg, int kend,
                                      const bool preA = false, const bool preB = false) {
    if constexpr (DEEP) {
        xps_big::BigStage32& st = *reinterpret_cast<xps_big::BigStage32*>(big_smem);
        xps_big::BigLoader32<AK> la;
        xps_big::BigLoader32<BK> lb;
        la.init(A, lda, m0, kbeg, threadIdx.x);
        lb.init(B, ldb, n0, kbeg, threadIdx.x);
        xps_big::big_pipeline32<AK, BK>(acc, csum, want_csum, la, lb, (kend - kbeg) / 32, st, preA, preB);
    } else {
        xps_big::BigStage& st = *reinterpret_cast<xps_big::BigStage*>(big_smem);
        xps_big::BigLoader<AK> la;
        xps_big::BigLoader<BK> lb;
        la.init(A, lda, m0, kbeg, threadIdx.x);
        lb.init(B, ldb, n0, kbeg, threadIdx.x);
        if constexpr (FMT >= 0) xps_big::big_pipeline_t<AK, BK, (FMT & 1) != 0, (FMT & 2) != 0>(acc, csum, want_csum, la, lb, (kend - kbeg) / BKT, st);
        else xps_big::big_pipeline<AK, BK>(acc, csum, want_csum, la, lb, (kend - kbeg) / BKT, st, preA, preB);
    }
}

template <bool AK, bool BK, bool DEEP, int FMT = 0>
__global__ __launch_bounds__(512, 1) void gemm_big_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ B,
                                                           long long ldb, const float* __restrict__ A2, const float* __restrict__ B2,
                                                           int K2, float* __restrict__ C, long long ldc,
                                                           const float* __restrict__ bias, int N, int K, int accumulate, int fmt) {
    // FMT == 5: the FMT == 4 loop on a SKINNY problem -- N < 256 columns in one 256-wide tile: B's rows are readable to 256
    // columns (its leading dimension says so; what lies beyond column N only reaches output columns that are never stored), the
    // waves whose 64 columns lie beyond N idle in the MFMA phase, C (leading dimension ldc) gets its N columns
    const int tiles_n = FMT == 5 ? 1 : N / xps_big::TN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * xps_big::TM, n0 = (lid % tiles_n) * xps_big::TN;
    f32x16 acc[4][2];
    xps_big::big_zero(acc);
    f32x4 nocs = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FMT == 5) {
        const bool act = (int)((threadIdx.x >> 6) & 3) * 64 < N;
        xps_big::direct_dma_pipeline<!BK>(acc, A, lda, B, ldb, m0, 0, 0, K / 32, big_smem, act);
        if (A2) xps_big::direct_dma_pipeline<!BK>(acc, A2, lda, B2, ldb, m0, 0, 0, K2 / 32, big_smem, act);
        xps_big::big_store_c_masked(acc, C, ldc, bias, m0, N, accumulate);
        return;
    }
    // fmt: bit 0 = A (and A2), bit 1 = B (and B2) are XPS_FMT_SPLIT4 operands; 16-deep stages: fmt == FMT (template), two
    // k loops in the kernel instead of eight (which spilled 300-400 registers)
    // (32-deep stages, opt-in: fp32 operands only -- the host never pairs them with split4 operands)
    // FMT == 4: both operands XPS_FMT_SPLIT4, A stored [m][k], k ranges multiples of 32: the LDS-DMA k loop (xps_gemm_dma.h)
    if constexpr (FMT == 4) {
        xps_big::direct_dma_pipeline<!BK>(acc, A, lda, B, ldb, m0, n0, 0, K / 32, big_smem);
        if (A2) xps_big::direct_dma_pipeline<!BK>(acc, A2, lda, B2, ldb, m0, n0, 0, K2 / 32, big_smem);
    } else {
        big_accumulate<AK, BK, DEEP, DEEP ? -1 : FMT>(acc, nocs, false, A, lda, B, ldb, m0, n0, 0, K, !DEEP && (fmt & 1) != 0, !DEEP && (fmt & 2) != 0);
        if (A2) big_accumulate<AK, BK, DEEP, DEEP ? -1 : FMT>(acc, nocs, false, A2, lda, B2, ldb, m0, n0, 0, K2, !DEEP && (fmt & 1) != 0, !DEEP && (fmt & 2) != 0);
    }
    xps_big::big_store_c(acc, C, ldc, bias, m0, n0, accumulate);
}

// same A, up to 4 (B, bias, C): logical order (m-tile, problem, n-tile)
template <bool DEEP, bool DMA = false>
__global__ __launch_bounds__(512, 1) void gemm_big_nt_multi_kernel(const float* __restrict__ A, long long lda, NtMulti pm, long long ldb,
                                                                    long long ldc, int N, int K, int nprob, int fmt) {
    const int tiles_n = N / xps_big::TN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_m = nprob * tiles_n, rem = lid % per_m;
    const int z = rem / tiles_n;
    const int m0 = (lid / per_m) * xps_big::TM, n0 = (rem % tiles_n) * xps_big::TN;
    f32x16 acc[4][2];
    xps_big::big_zero(acc);
    f32x4 nocs = {0.f, 0.f, 0.f, 0.f};
    if constexpr (DMA) xps_big::direct_dma_pipeline<false>(acc, A, lda, pm.B[z], ldb, m0, n0, 0, K / 32, big_smem);     // (both operands split4)
    else big_accumulate<true, true, DEEP>(acc, nocs, false, A, lda, pm.B[z], ldb, m0, n0, 0, K, !DEEP && (fmt & 1) != 0, !DEEP && (fmt & 2) != 0);
    xps_big::big_store_c(acc, pm.C[z], ldc, pm.bias[z], m0, n0, 0);
}

// grouped weight gradients: the problems of a TnGroup marked `big`; block = (problem, k-split, 256 x 256 tile); slabs in
// the layout of the 128 x 128 kernel (four sub-tiles per block) so that gemm_tn_grouped_reduce serves both launches
template <bool DEEP>
__global__ __launch_bounds__(512, 1) void gemm_big_tn_kernel(TnGroup g, float* __restrict__ ws) {
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    int pi = -1;
#pragma unroll 1
    for (int i = 0; i < g.n; ++i)
        if (g.p[i].big && lid >= g.p[i].block_start) pi = i;
    const TnProb& P = g.p[pi];
    const int local = lid - P.block_start;
    const int tiles_n = P.N / xps_big::TN, ntiles = (P.M / xps_big::TM) * tiles_n;
    const int tile = local % ntiles, z = local / ntiles;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int kbeg = z * P.kchunk, kend = min(P.K, kbeg + P.kchunk);
    const long long ntiles128 = (long long)(P.M / BM) * P.tiles_n;
    const long long split_stride = tn_split_stride(ntiles128, P.M, P.colsum != nullptr);
    float* slab = ws + P.slab_off + (long long)z * split_stride;
    const bool want_cs = P.colsum && tn == 0;
    f32x16 acc[4][2];
    xps_big::big_zero(acc);
    if (!DEEP && P.big == 2) {
        // both operands carry the hi / lo split: LDS-DMA k loop (no staging registers, no split arithmetic, no ds_write); the
        // column sums ride on the matrix pipe.  Same products, bit for bit, as the register-staged loop below.
        // column sums: the n-tile blocks of an A tile share its eight 32-column groups (block tn: groups [cs_lo, cs_hi))
        f32x16 cacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) cacc[r] = 0.f;
        const int per = (8 + tiles_n - 1) / tiles_n;
        const int cs_lo = min(8, tn * per), cs_hi = min(8, cs_lo + per);
        const bool cs_here = P.colsum && cs_lo < cs_hi;
        if (cs_here) xps_big::tn_dma_pipeline<true>(acc, cacc, P.A, P.ra.ld, P.B, P.rb.ld, tm * xps_big::TM, tn * xps_big::TN, kbeg, (kend - kbeg) / BKT, big_smem, cs_lo, cs_hi);
        else xps_big::tn_dma_pipeline<false>(acc, cacc, P.A, P.ra.ld, P.B, P.rb.ld, tm * xps_big::TM, tn * xps_big::TN, kbeg, (kend - kbeg) / BKT, big_smem);
        float* subd[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
            subd[s] = slab + ((long long)(2 * tm + (s >> 1)) * P.tiles_n + 2 * tn + (s & 1)) * TN_TILE;
        xps_big::big_slab_store(acc, subd);
        if (cs_here) xps_big::dma_colsum_store(cacc, slab + ntiles128 * TN_TILE + tm * xps_big::TM, cs_lo, cs_hi);
        return;
    }
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    big_accumulate<false, false, DEEP>(acc, csum, want_cs, P.A, P.ra.ld, P.B, P.rb.ld, tm * xps_big::TM, tn * xps_big::TN, kbeg, kend,
                                       !DEEP && (P.vecA & 2) != 0, !DEEP && (P.vecB & 2) != 0);
    float* sub[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        sub[s] = slab + ((long long)(2 * tm + (s >> 1)) * P.tiles_n + 2 * tn + (s & 1)) * TN_TILE;
    xps_big::big_slab_store(acc, sub);
    if (want_cs) {
        // thread (k rows tid / 64 + {0, 8}, x group tid % 64) holds the sums of columns 4 xg .. 4 xg + 3 over its k rows:
        // the eight waves are folded through LDS in a fixed order (big_pipeline ends with a barrier: the stage is free)
        float* red = reinterpret_cast<float*>(big_smem);
        const int tid = threadIdx.x;
        *reinterpret_cast<f32x4*>(&red[(tid >> 6) * 256 + (tid & 63) * 4]) = csum;
        __syncthreads();
        if (tid < 256) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[w * 256 + tid];
            slab[ntiles128 * TN_TILE + tm * xps_big::TM + tid] = v;
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, long long slab_stride,
                                                            float* __restrict__ C, RowMap rc, int M, int N, int accumulate) {
    __shared__ float part[4][RED_OUT];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long idx = (long long)blockIdx.x * RED_OUT + l;
    const bool live = idx < (long long)M * N;
    part[w][l] = live ? slab_sum(slabs + idx, slab_stride, splits, w) : 0.f;
    __syncthreads();
    if (w == 0 && live) {
        float s = (part[0][l] + part[1][l]) + (part[2][l] + part[3][l]);
        float* p = C + rc.off((int)(idx / N)) + (int)(idx % N);
        if (accumulate) s += *p;
        *p = s;
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool map_vec_ok(const float* p, const RowMap& r) {
    return aligned16(p) && (r.ld % 4 == 0) && (r.gs % 4 == 0);
}
// XPS_FMT_SPLIT4 input operand (xps.h): 2 = to be OR-ed into the kernel's vec flag, 0 = plain fp32, -1 = not servable
// (needs whole, aligned 16-byte groups along the contiguous index: `extent` = its length; bf16x3 mode is checked by the caller)
inline int split4_flag(const float* p, const RowMap& r, long long extent) {
    if (r.fmt == 0) return 0;
    if (r.fmt != 1 || !map_vec_ok(p, r) || extent % 4 != 0) return -1;
    return 2;
}

int tn_splits(int M, int N, int K) {
    long long tiles = (long long)cdiv(M, BM) * cdiv(N, BN);
    int want = (int)((768 + tiles - 1) / tiles);
    int maxs = cdiv(K, 256);
    int s = want < maxs ? want : maxs;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return s;
}

}  // namespace

namespace {
// Product precision of the tile kernels: 0 = fp32 MFMA (exact fp32 fma chains), 1 = bf16 split products (3 bf16 MFMAs
// per product, ~2^-16 relative product error, fp32 accumulate).  Process-wide; XPS_GEMM_PRECISION=fp32|bf16x3 sets the
// initial value, xps_set_gemm_precision() changes it (not while launches of another thread are in flight).
std::atomic<int>& gemm_mode() {
    static std::atomic<int> mode([] {
        const char* e = getenv("XPS_GEMM_PRECISION");
        if (e && (!strcmp(e, "fp32") || !strcmp(e, "f32") || !strcmp(e, "0"))) return 0;
        if (e && (!strcmp(e, "bf16x3") || !strcmp(e, "1"))) return 1;
        return XPS_GEMM_DEFAULT_MODE;
    }());
    return mode;
}
inline bool bf_mode() { return gemm_mode().load(std::memory_order_relaxed) == 1; }

// 64-row tiles when 128-row tiles would leave the chip under-filled (< ~2 blocks per CU)
// (long contractions -- the rows a 256-tile launch leaves behind, K >= 1024 -- are issue-bound, not HBM-bound: there the 128-row
//  tile's better MFMA-per-staged-element ratio wins as soon as every CU gets two blocks)
inline bool use_small_tiles(int M, int N, int K = 0) {
    static const int thr = [] {
        const char* e = getenv("XPS_GEMM_SMALL_TILE_BLOCKS");
        int v = e ? atoi(e) : -1;
        return v >= 0 ? v : 2048;
    }();
    const long long n128 = (long long)cdiv(M, 128) * cdiv(N, BN);
    static const bool longk = [] { const char* e = getenv("XPS_GEMM_LONGK_128"); return !(e && e[0] == '0'); }();
    if (longk && K >= 1024 && n128 >= 512) return false;
    return M > 64 && n128 < thr;
}

// 256 x 256 tiles (xps_gemm_big.h): bf16 split-product mode, whole tiles only, plain 16-byte aligned operands, and
// enough tiles to give every CU one.  XPS_GEMM_BIG=0 switches the path off (A/B runs, tests of the small-tile kernels).
std::atomic<int>& big_switch() {
    static std::atomic<int> on([] { const char* e = getenv("XPS_GEMM_BIG"); return (e && e[0] == '0') ? 0 : 1; }());
    return on;
}
inline bool big_enabled() { return big_switch().load(std::memory_order_relaxed) != 0 && bf_mode(); }
// LDS-DMA k loop of the 256-tile weight-gradient kernel (xps_gemm_dma.h); XPS_GEMM_DMA=0 (read per call: A/B tests toggle it)
inline bool dma_enabled() { const char* e = getenv("XPS_GEMM_DMA"); return !(e && e[0] == '0'); }
// weight-stationary projection kernel (proj_ws_kernel): XPS_PROJ_WS=0 keeps the tile kernels (read per call: A/B tests toggle it)
inline bool proj_ws_enabled() { const char* e = getenv("XPS_PROJ_WS"); return !(e && e[0] == '0'); }
inline bool big_plain(const float* p, const RowMap& r, int rows) { return aligned16(p) && r.ld % 4 == 0 && r.rpg >= rows; }
constexpr int BIG_LDS = (int)sizeof(xps_big::BigStage), BIG_LDS32 = (int)sizeof(xps_big::BigStage32);
constexpr int BIG_TN_LDS = BIG_LDS > xps_big::DMA_LDS ? BIG_LDS : xps_big::DMA_LDS;      // the weight-gradient kernel holds both k loops
template <typename Kern>
inline bool big_prepare(Kern kern, int bytes) {
    // once per kernel: allow the 96- / 144-KB dynamic LDS block
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}
// 32-deep stages (opt-in: XPS_GEMM_BIG_DEEP=1, and only where every k range is a multiple of 32).  Measured: the single
// launches gain (projection 418 -> 396 us, input gradient 423 -> 392 us, 8192^3 0.40 -> 0.45 of peak; the TN form loses
// 1-5 %), the configs[3] step does not (8.26-8.27 vs 8.20-8.23 ms, A/B on one box): 16-deep stays the default.
inline bool big_deep_allowed() {
    static const bool on = [] { const char* e = getenv("XPS_GEMM_BIG_DEEP"); return e && e[0] == '1'; }();
    return on;
}
inline int big_min_tiles() {
    static const int v = [] { const char* e = getenv("XPS_GEMM_BIG_MIN_TILES"); int x = e ? atoi(e) : 0; return x > 0 ? x : 192; }();
    return v;
}

inline int device_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) return n;
        return 256;
    }();
    return cus;
}
// Tile-count quantisation of the one-block-per-CU 256-tile kernels: 640 tiles on 256 CUs are 2.5 rounds that cost 3.  When
// the last round would be at most three quarters full, the 256-tile launch takes the leading row tiles that fill whole
// rounds and the trailing rows go to the 128-tile kernels (three blocks per CU: their round is a fraction of a 256-tile
// round) -- same bits either way (both tile shapes accumulate a product whose k range is not split in the same order).
// Returns the number of leading ROWS for the 256-tile launch (== M: no split).  XPS_GEMM_BIG_TAIL=0: never split.
inline int big_split_rows(int M, int tiles_per_row_tile) {
    static const bool on = [] { const char* e = getenv("XPS_GEMM_BIG_TAIL"); return !(e && e[0] == '0'); }();
    const int cus = device_cus();
    const long long mt = M / xps_big::TM, total = mt * tiles_per_row_tile;
    const long long rounds = total / cus, rem = total % cus;
    if (!on || rounds < 1 || rem == 0 || rem * 4 > (long long)cus * 3) return M;
    const long long mb = (rounds * cus) / tiles_per_row_tile;
    if (mb < 1 || mb >= mt) return M;
    return (int)(mb * xps_big::TM);
}

template <bool AK, bool BK>
int launch_gemm(const float* A, const RowMap& ra, const float* B, const RowMap& rb, const float* A2, const float* B2, int K2,
                float* C, const RowMap& rc, const float* bias, int M, int N, int K, int accumulate, hipStream_t st) {
    const int kchunk = ((K + BKT - 1) / BKT) * BKT + BKT;
    int vecA = (int)(map_vec_ok(A, ra) && (!A2 || map_vec_ok(A2, ra)));
    int vecB = (int)(map_vec_ok(B, rb) && (!B2 || map_vec_ok(B2, rb)));
    // XPS_FMT_SPLIT4 operands: bit 1 of the vec flags (-2: the request cannot be served)
    int fA = split4_flag(A, ra, AK ? K : M), fB = split4_flag(B, rb, BK ? K : N);
    if (A2 && fA > 0) fA = split4_flag(A2, ra, AK ? K2 : M);
    if (B2 && fB > 0) fB = split4_flag(B2, rb, BK ? K2 : N);
    if (fA < 0 || fB < 0 || ((fA | fB) && (!bf_mode() || M <= SM_MAXM))) return -2;
    vecA |= fA; vecB |= fB;
    const int fmt = (fA ? 1 : 0) | (fB ? 2 : 0);
    if (M <= SM_MAXM) {
        hipLaunchKernelGGL((gemm_small_kernel<AK, BK>), dim3(cdiv(N, 64)), dim3(1024), 0, st, A, ra, B, rb, A2, B2, K2, C, rc,
                           bias, M, N, K, accumulate, vecB);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    if constexpr (AK && !BK) {
        // skinny input gradient (configs[3] layer 0: 40960 x 100 x 3072, HBM-bound on A): ONE 256-wide tile per 256 rows on the
        // LDS-DMA loop; needs B's rows readable (and split4) to 256 columns: the caller hands a zero-padded image (rb.ld >= 256)
        if (big_enabled() && dma_enabled() && fmt == 3 && N < xps_big::TN && N % 4 == 0 && rb.ld >= xps_big::TN && M % xps_big::TM == 0 &&
            M / xps_big::TM >= 96 && K % 32 == 0 && K2 % 32 == 0 && K + K2 >= 512 && big_plain(A, ra, M) && big_plain(B, rb, K) &&
            (!A2 || (big_plain(A2, ra, M) && big_plain(B2, rb, K2))) && rc.rpg >= M) {
            constexpr int lds = xps_big::direct_dma_lds<true>();
            static const bool ready5 = big_prepare(gemm_big_kernel<true, false, false, 5>, lds);
            if (ready5) {
                hipLaunchKernelGGL((gemm_big_kernel<true, false, false, 5>), dim3(M / xps_big::TM), dim3(xps_big::NTHR), lds, st,
                                   A, ra.ld, B, rb.ld, A2, B2, K2, C, rc.ld, bias, N, K, accumulate, fmt);
                return hipGetLastError() == hipSuccess ? 0 : -1;
            }
        }
    }
    if (big_enabled() && M % xps_big::TM == 0 && N % xps_big::TN == 0 && K % BKT == 0 && K2 % BKT == 0 && K >= 64 &&
        (long long)(M / xps_big::TM) * (N / xps_big::TN) >= big_min_tiles() &&
        big_plain(A, ra, AK ? M : K) && big_plain(B, rb, BK ? N : K) && (!A2 || (big_plain(A2, ra, AK ? M : K2) && big_plain(B2, rb, BK ? N : K2))) &&
        rc.rpg >= M) {
        static const bool ready = big_prepare(gemm_big_kernel<AK, BK, false, 0>, BIG_LDS) && big_prepare(gemm_big_kernel<AK, BK, false, 1>, BIG_LDS) &&
                                  big_prepare(gemm_big_kernel<AK, BK, false, 2>, BIG_LDS) && big_prepare(gemm_big_kernel<AK, BK, false, 3>, BIG_LDS) &&
                                  big_prepare(gemm_big_kernel<AK, BK, true>, BIG_LDS32);
        if (ready) {
            const int Mfull = M;
            M = big_split_rows(M, N / xps_big::TN);
            const dim3 bgrid((M / xps_big::TM) * (N / xps_big::TN));
            // the rows behind the whole rounds: 128-tile kernels (the recursion ends there: too few tiles for another 256-tile launch)
            const int Mtail = Mfull - M;
            auto run_tail = [&]() -> int {
                if (Mtail <= 0) return 0;
                const float* At = AK ? A + (long long)M * ra.ld : A + M;
                const float* A2t = A2 ? (AK ? A2 + (long long)M * ra.ld : A2 + M) : nullptr;
                return launch_gemm<AK, BK>(At, ra, B, rb, A2t, B2, K2, C + (long long)M * rc.ld, rc, bias, Mtail, N, K, accumulate, st);
            };
            bool dma_done = false;
            if constexpr (AK) {
                // both operands split4, A stored [m][k], whole 32-deep stages: the LDS-DMA k loop (same bits; XPS_GEMM_DMA=0: never)
                if (fmt == 3 && K % 32 == 0 && K2 % 32 == 0 && dma_enabled()) {
                    constexpr int lds = xps_big::direct_dma_lds<!BK>();
                    static const bool ready4 = big_prepare(gemm_big_kernel<AK, BK, false, 4>, lds);
                    if (ready4) {
                        hipLaunchKernelGGL((gemm_big_kernel<AK, BK, false, 4>), bgrid, dim3(xps_big::NTHR), lds, st,
                                           A, ra.ld, B, rb.ld, A2, B2, K2, C, rc.ld, bias, N, K, accumulate, fmt);
                        dma_done = true;
                    }
                }
            }
            if (dma_done) {
            } else if (big_deep_allowed() && K % 32 == 0 && K2 % 32 == 0 && fmt == 0)
                hipLaunchKernelGGL((gemm_big_kernel<AK, BK, true>), bgrid, dim3(xps_big::NTHR), BIG_LDS32, st,
                                   A, ra.ld, B, rb.ld, A2, B2, K2, C, rc.ld, bias, N, K, accumulate, fmt);
            else if (fmt == 0)
                hipLaunchKernelGGL((gemm_big_kernel<AK, BK, false, 0>), bgrid, dim3(xps_big::NTHR), BIG_LDS, st,
                                   A, ra.ld, B, rb.ld, A2, B2, K2, C, rc.ld, bias, N, K, accumulate, fmt);
            else if (fmt == 1)
                hipLaunchKernelGGL((gemm_big_kernel<AK, BK, false, 1>), bgrid, dim3(xps_big::NTHR), BIG_LDS, st,
                                   A, ra.ld, B, rb.ld, A2, B2, K2, C, rc.ld, bias, N, K, accumulate, fmt);
            else if (fmt == 2)
                hipLaunchKernelGGL((gemm_big_kernel<AK, BK, false, 2>), bgrid, dim3(xps_big::NTHR), BIG_LDS, st,
                                   A, ra.ld, B, rb.ld, A2, B2, K2, C, rc.ld, bias, N, K, accumulate, fmt);
            else
                hipLaunchKernelGGL((gemm_big_kernel<AK, BK, false, 3>), bgrid, dim3(xps_big::NTHR), BIG_LDS, st,
                                   A, ra.ld, B, rb.ld, A2, B2, K2, C, rc.ld, bias, N, K, accumulate, fmt);
            if (hipGetLastError() != hipSuccess) return -1;
            return run_tail();
        }
    }
#define XPS_LAUNCH_GEMM(MI_, EDGE_, BF_)                                                                              \
    do {                                                                                                              \
        if (BF_ && fmt)                                                                                               \
            hipLaunchKernelGGL((gemm_f32_kernel<AK, BK, MI_, EDGE_, BF_, BF_>), grid, dim3(256), 0, st, A, ra, B, rb, A2, B2, K2, C, rc, \
                               bias, M, N, K, kchunk, 0LL, accumulate, vecA, vecB);                                   \
        else                                                                                                          \
            hipLaunchKernelGGL((gemm_f32_kernel<AK, BK, MI_, EDGE_, BF_>), grid, dim3(256), 0, st, A, ra, B, rb, A2, B2, K2, C, rc, \
                               bias, M, N, K, kchunk, 0LL, accumulate, vecA, vecB);                                   \
    } while (0)
    const bool bf = bf_mode();
    if (use_small_tiles(M, N, K + K2)) {
        dim3 grid(cdiv(N, BN) * cdiv(M, 64));
        const bool edge = (M % 64) || (N % BN);
        if (bf) { if (edge) XPS_LAUNCH_GEMM(1, true, true); else XPS_LAUNCH_GEMM(1, false, true); }
        else { if (edge) XPS_LAUNCH_GEMM(1, true, false); else XPS_LAUNCH_GEMM(1, false, false); }
    } else {
        dim3 grid(cdiv(N, BN) * cdiv(M, 128));
        const bool edge = (M % 128) || (N % BN);
        if (bf) { if (edge) XPS_LAUNCH_GEMM(2, true, true); else XPS_LAUNCH_GEMM(2, false, true); }
        else { if (edge) XPS_LAUNCH_GEMM(2, true, false); else XPS_LAUNCH_GEMM(2, false, false); }
    }
#undef XPS_LAUNCH_GEMM
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
}  // namespace

extern "C" int xps_gemm_nt_f32(const float* A, const xps_rowmap* ra_, const float* B, const xps_rowmap* rb_,
                               float* C, const xps_rowmap* rc_, const float* bias,
                               int M, int N, int K, int accumulate, void* stream) {
    XPS_CHECK_ARG(A && B && C && ra_ && rb_ && rc_, "null argument");
    XPS_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative size");
    if (M == 0 || N == 0) return XPS_OK;
    RowMap ra = to_rowmap(ra_), rb = to_rowmap(rb_), rc = to_rowmap(rc_);
    const int lrc = launch_gemm<true, true>(A, ra, B, rb, nullptr, nullptr, 0, C, rc, bias, M, N, K, accumulate, (hipStream_t)stream);
    if (lrc == -2) {
        xps_set_error("xps_gemm_nt_f32: XPS_FMT_SPLIT4 operand needs bf16x3 mode, M > 16, 16-byte aligned rows and a contiguous extent that is a multiple of 4");
        return XPS_E_INVALID;
    }
    if (lrc) {
        xps_set_error("xps_gemm_nt_f32: launch failed: %s", hipGetErrorString(hipGetLastError()));
        return XPS_E_HIP;
    }
    return XPS_OK;
}

extern "C" int xps_gemm_nn_f32(const float* A, const xps_rowmap* ra_, const float* B, const xps_rowmap* rb_,
                               float* C, const xps_rowmap* rc_,
                               int M, int N, int K, int accumulate, void* stream) {
    XPS_CHECK_ARG(A && B && C && ra_ && rb_ && rc_, "null argument");
    XPS_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative size");
    if (M == 0 || N == 0) return XPS_OK;
    RowMap ra = to_rowmap(ra_), rb = to_rowmap(rb_), rc = to_rowmap(rc_);
    const int lrc = launch_gemm<true, false>(A, ra, B, rb, nullptr, nullptr, 0, C, rc, nullptr, M, N, K, accumulate, (hipStream_t)stream);
    if (lrc == -2) {
        xps_set_error("xps_gemm_nn_f32: XPS_FMT_SPLIT4 operand needs bf16x3 mode, M > 16, 16-byte aligned rows and a contiguous extent that is a multiple of 4");
        return XPS_E_INVALID;
    }
    if (lrc) {
        xps_set_error("xps_gemm_nn_f32: launch failed: %s", hipGetErrorString(hipGetLastError()));
        return XPS_E_HIP;
    }
    return XPS_OK;
}

extern "C" int xps_gemm_nt_multi_f32(const float* A, const xps_rowmap* ra_, const float* const* B, const xps_rowmap* rb_,
                                     float* const* C, const xps_rowmap* rc_, const float* const* bias, int nprob,
                                     int M, int N, int K, void* stream) {
    XPS_CHECK_ARG(A && B && C && ra_ && rb_ && rc_, "null argument");
    XPS_CHECK_ARG(nprob >= 1 && nprob <= 4, "1..4 problems");
    XPS_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative size");
    if (M == 0 || N == 0) return XPS_OK;
    RowMap ra = to_rowmap(ra_), rb = to_rowmap(rb_), rc = to_rowmap(rc_);
    NtMulti pm;
    bool vb = true;
    for (int i = 0; i < 4; ++i) {
        const int j = i < nprob ? i : 0;
        XPS_CHECK_ARG(B[j] && C[j], "null problem pointer");
        pm.B[i] = B[j]; pm.C[i] = C[j]; pm.bias[i] = bias ? bias[j] : nullptr;
        vb = vb && map_vec_ok(B[j], rb);
    }
    int vecA = (int)map_vec_ok(A, ra), vecB = (int)vb;
    int fA = split4_flag(A, ra, K), fB = 0;
    for (int i = 0; i < nprob && fB >= 0; ++i) fB = split4_flag(B[i], rb, K);
    if (fA < 0 || fB < 0 || ((fA | fB) && !bf_mode())) {
        xps_set_error("xps_gemm_nt_multi_f32: XPS_FMT_SPLIT4 operand needs bf16x3 mode, 16-byte aligned rows and K %% 4 == 0");
        return XPS_E_INVALID;
    }
    vecA |= fA; vecB |= fB;
    const int fmt = (fA ? 1 : 0) | (fB ? 2 : 0);
    // weight-stationary projection: small K (the weights of a 256-column slice fit the registers of a block), many rows
    if (bf_mode() && proj_ws_enabled() && K >= 32 && K <= 256 && K % 4 == 0 && N % 32 == 0 && (long long)nprob * N >= 256 && M >= 4096 &&
        vecA && vb && ra.rpg >= M && rb.rpg >= N && rc.rpg >= M) {
        const int KT = 2 * ((K + 31) / 32);
        static const int cus = [] {                          // one persistent block per CU
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) return n;
            return 256;
        }();
        const int S = cdiv((long long)nprob * N, 256), tiles = cdiv(M, 64);
        int grid = cus < tiles * S ? cus : tiles * S;
        grid = (grid / S) * S;                              // every row group has all its slices
        if (grid >= S) {
#define XPS_LAUNCH_WS(KT_)                                                                                                            \
    do {                                                                                                                              \
        static const bool ok_ = big_prepare(proj_ws_kernel<KT_>, proj_ws_lds(KT_));                                                    \
        if (!ok_) { xps_set_error("xps_gemm_nt_multi_f32: cannot reserve %d bytes of LDS", proj_ws_lds(KT_)); return XPS_E_HIP; }       \
        hipLaunchKernelGGL(proj_ws_kernel<KT_>, dim3(grid), dim3(512), proj_ws_lds(KT_), (hipStream_t)stream, A, ra.ld, pm, rb.ld, rc.ld, \
                           M, N, K, nprob, fA ? 1 : 0, fB ? 1 : 0);                                                                      \
    } while (0)
            switch (KT) {
                case 2: XPS_LAUNCH_WS(2); break;
                case 4: XPS_LAUNCH_WS(4); break;
                case 6: XPS_LAUNCH_WS(6); break;
                case 8: XPS_LAUNCH_WS(8); break;
                case 10: XPS_LAUNCH_WS(10); break;
                case 12: XPS_LAUNCH_WS(12); break;
                case 14: XPS_LAUNCH_WS(14); break;
                default: XPS_LAUNCH_WS(16); break;
            }
#undef XPS_LAUNCH_WS
            XPS_CHECK_LAUNCH();
            return XPS_OK;
        }
    }
    if (big_enabled() && M % xps_big::TM == 0 && N % xps_big::TN == 0 && K % BKT == 0 && K >= 64 && vb &&
        (long long)(M / xps_big::TM) * (N / xps_big::TN) * nprob >= big_min_tiles() &&
        big_plain(A, ra, M) && rb.rpg >= N && rc.rpg >= M) {
        static const bool ready = big_prepare(gemm_big_nt_multi_kernel<false>, BIG_LDS) && big_prepare(gemm_big_nt_multi_kernel<true>, BIG_LDS32);
        if (ready) {
            const int Mfull = M;
            M = big_split_rows(M, (N / xps_big::TN) * nprob);
            const dim3 bgrid((M / xps_big::TM) * (N / xps_big::TN) * nprob);
            static const bool ready_dma = big_prepare(gemm_big_nt_multi_kernel<false, true>, xps_big::direct_dma_lds<false>());
            if (fmt == 3 && K % 32 == 0 && dma_enabled() && ready_dma)          // both operands split4: the LDS-DMA k loop (same bits)
                hipLaunchKernelGGL((gemm_big_nt_multi_kernel<false, true>), bgrid, dim3(xps_big::NTHR), xps_big::direct_dma_lds<false>(),
                                   (hipStream_t)stream, A, ra.ld, pm, rb.ld, rc.ld, N, K, nprob, fmt);
            else if (big_deep_allowed() && K % 32 == 0 && fmt == 0)
                hipLaunchKernelGGL(gemm_big_nt_multi_kernel<true>, bgrid, dim3(xps_big::NTHR), BIG_LDS32, (hipStream_t)stream, A, ra.ld, pm,
                                   rb.ld, rc.ld, N, K, nprob, fmt);
            else
                hipLaunchKernelGGL(gemm_big_nt_multi_kernel<false>, bgrid, dim3(xps_big::NTHR), BIG_LDS,
                               (hipStream_t)stream, A, ra.ld, pm, rb.ld, rc.ld, N, K, nprob, fmt);
            XPS_CHECK_LAUNCH();
            if (Mfull > M) {
                // the rows behind the whole rounds: the 128-tile kernels (see big_split_rows)
                float* Ct[4];
                for (int i = 0; i < nprob; ++i) Ct[i] = C[i] + (long long)M * rc.ld;
                return xps_gemm_nt_multi_f32(A + (long long)M * ra.ld, ra_, B, rb_, Ct, rc_, bias, nprob, Mfull - M, N, K, stream);
            }
            return XPS_OK;
        }
    }
#define XPS_LAUNCH_NTM(MI_, EDGE_, BF_)                                                                                  \
    do {                                                                                                                 \
        if (BF_ && fmt)                                                                                                  \
            hipLaunchKernelGGL((gemm_nt_multi_kernel<MI_, EDGE_, BF_, BF_>), dim3(cdiv(N, BN) * cdiv(M, 64 * MI_) * nprob), dim3(256), 0, \
                               (hipStream_t)stream, A, ra, pm, rb, rc, M, N, K, nprob, vecA, vecB);                        \
        else                                                                                                             \
            hipLaunchKernelGGL((gemm_nt_multi_kernel<MI_, EDGE_, BF_>), dim3(cdiv(N, BN) * cdiv(M, 64 * MI_) * nprob), dim3(256), 0,    \
                               (hipStream_t)stream, A, ra, pm, rb, rc, M, N, K, nprob, vecA, vecB);                        \
    } while (0)
    const bool bf = bf_mode();
    if (use_small_tiles(M, N * nprob, K)) {
        const bool edge = (M % 64) || (N % BN);
        if (bf) { if (edge) XPS_LAUNCH_NTM(1, true, true); else XPS_LAUNCH_NTM(1, false, true); }
        else { if (edge) XPS_LAUNCH_NTM(1, true, false); else XPS_LAUNCH_NTM(1, false, false); }
    } else {
        const bool edge = (M % 128) || (N % BN);
        if (bf) { if (edge) XPS_LAUNCH_NTM(2, true, true); else XPS_LAUNCH_NTM(2, false, true); }
        else { if (edge) XPS_LAUNCH_NTM(2, true, false); else XPS_LAUNCH_NTM(2, false, false); }
    }
#undef XPS_LAUNCH_NTM
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

extern "C" int xps_gemm_nn2_f32(const float* A1, const float* B1, int K1, const float* A2, const float* B2, int K2,
                                const xps_rowmap* ra_, const xps_rowmap* rb_, float* C, const xps_rowmap* rc_,
                                int M, int N, int accumulate, void* stream) {
    XPS_CHECK_ARG(A1 && B1 && A2 && B2 && C && ra_ && rb_ && rc_, "null argument");
    XPS_CHECK_ARG(M >= 0 && N >= 0 && K1 >= 0 && K2 >= 0, "negative size");
    if (M == 0 || N == 0) return XPS_OK;
    RowMap ra = to_rowmap(ra_), rb = to_rowmap(rb_), rc = to_rowmap(rc_);
    const int lrc = launch_gemm<true, false>(A1, ra, B1, rb, A2, B2, K2, C, rc, nullptr, M, N, K1, accumulate, (hipStream_t)stream);
    if (lrc == -2) {
        xps_set_error("xps_gemm_nn2_f32: XPS_FMT_SPLIT4 operand needs bf16x3 mode, M > 16, 16-byte aligned rows and a contiguous extent that is a multiple of 4");
        return XPS_E_INVALID;
    }
    if (lrc) {
        xps_set_error("xps_gemm_nn2_f32: launch failed: %s", hipGetErrorString(hipGetLastError()));
        return XPS_E_HIP;
    }
    return XPS_OK;
}

extern "C" size_t xps_gemm_tn_f32_workspace(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 16;
    int s = tn_splits(M, N, K);
    return (size_t)s * (size_t)M * (size_t)N * sizeof(float) + 16;
}

extern "C" int xps_gemm_tn_f32(const float* A, const xps_rowmap* ra_, const float* B, const xps_rowmap* rb_,
                               float* C, const xps_rowmap* rc_,
                               int M, int N, int K, int accumulate,
                               void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(A && B && C && ra_ && rb_ && rc_, "null argument");
    XPS_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "negative size");
    if (M == 0 || N == 0) return XPS_OK;
    RowMap ra = to_rowmap(ra_), rb = to_rowmap(rb_), rc = to_rowmap(rc_);
    XPS_CHECK_ARG(ra.fmt == 0 && rb.fmt == 0, "XPS_FMT_SPLIT4 operands: use xps_gemm_tn_grouped_f32");
    const int splits = (K > 0) ? tn_splits(M, N, K) : 1;
    if (workspace_bytes < xps_gemm_tn_f32_workspace(M, N, K) || !workspace) {
        xps_set_error("xps_gemm_tn_f32: workspace too small (%zu < %zu)", workspace_bytes,
                      xps_gemm_tn_f32_workspace(M, N, K));
        return XPS_E_WORKSPACE;
    }
    XPS_CHECK_ARG(aligned16(workspace), "workspace must be 16-byte aligned");
    int kchunk = ((cdiv(K > 0 ? K : 1, splits) + BKT - 1) / BKT) * BKT;
    float* slabs = reinterpret_cast<float*>(workspace);
    RowMap rs;
    rs.gs = 0; rs.ld = N; rs.rpg = 1 << 30;
    const long long slab_stride = (long long)M * N;
    dim3 grid(cdiv(N, BN) * cdiv(M, BM) * splits);
    if (bf_mode())
        hipLaunchKernelGGL((gemm_f32_kernel<false, false, 2, true, true>), grid, dim3(256), 0, (hipStream_t)stream,
                           A, ra, B, rb, (const float*)nullptr, (const float*)nullptr, 0, slabs, rs, (const float*)nullptr,
                           M, N, K, kchunk, slab_stride, 0, (int)map_vec_ok(A, ra), (int)map_vec_ok(B, rb));
    else
        hipLaunchKernelGGL((gemm_f32_kernel<false, false, 2>), grid, dim3(256), 0, (hipStream_t)stream,
                           A, ra, B, rb, (const float*)nullptr, (const float*)nullptr, 0, slabs, rs, (const float*)nullptr,
                           M, N, K, kchunk, slab_stride, 0, (int)map_vec_ok(A, ra), (int)map_vec_ok(B, rb));
    XPS_CHECK_LAUNCH();
    long long total = (long long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(total, RED_OUT)), dim3(256), 0, (hipStream_t)stream,
                       slabs, splits, slab_stride, C, rc, M, N, accumulate);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

namespace {
int build_group(const xps_tn_problem* probs, int n, TnGroup& g, size_t& ws_floats, int big_choice = -1) {
    if (n < 1 || n > TN_MAXP) return -1;
    static const int target_blocks = [] {
        const char* e = getenv("XPS_TN_BLOCKS");
        int v = e ? atoi(e) : 0;
        return v > 0 ? v : 512;              // (768 until the flat reduce existed: 512 + flat is -5 % on the cfg-2 step, -1.7 % on configs[3])
    }();
    // problems the 256 x 256 kernel takes (whole tiles, plain aligned operands, a long k range); the others stay on
    // the 128 x 128 kernel.  Each class shares its own block budget among its problems in proportion to their tiles.
    const bool big_on = big_choice < 0 ? big_enabled() : big_choice != 0;      // (workspace query: both tilings)
    bool isbig[TN_MAXP];
    long long tiles_total = 0, tiles_big_total = 0;
    for (int i = 0; i < n; ++i) {
        const xps_tn_problem& q = probs[i];
        if (!q.A || !q.B || !q.C || q.M < 1 || q.N < 1 || q.K < 0) return -1;
        const RowMap ra = to_rowmap(&q.ra), rb = to_rowmap(&q.rb);
        isbig[i] = big_on && q.M % xps_big::TM == 0 && q.N % xps_big::TN == 0 && q.K % BKT == 0 && q.K >= 4096 &&
                   big_plain(q.A, ra, q.K) && big_plain(q.B, rb, q.K);
        if (isbig[i]) tiles_big_total += (long long)(q.M / xps_big::TM) * (q.N / xps_big::TN);
        else tiles_total += (long long)cdiv(q.M, BM) * cdiv(q.N, BN);
    }
    // big class: one block per CU and round; k-splits such that the launch is one or two nearly full rounds of 256 blocks
    int sp_big = 1;
    if (tiles_big_total > 0) {
        const int r1 = (int)(256 / tiles_big_total), r2 = (int)(512 / tiles_big_total);
        sp_big = (r1 >= 1 && r1 * tiles_big_total >= 230) ? r1 : (r2 >= 1 ? r2 : 1);
    }
    g.n = n;
    int blocks = 0, blocks_big = 0;
    long long off = 0, out = 0;
    for (int i = 0; i < n; ++i) {
        const xps_tn_problem& q = probs[i];
        TnProb& P = g.p[i];
        P.A = q.A; P.B = q.B; P.C = q.C; P.colsum = q.colsum_a;
        P.ra = to_rowmap(&q.ra); P.rb = to_rowmap(&q.rb); P.rc = to_rowmap(&q.rc);
        P.M = q.M; P.N = q.N; P.K = q.K; P.accumulate = q.accumulate;
        const int tiles = cdiv(q.M, BM) * cdiv(q.N, BN);
        const long long split_stride = tn_split_stride(tiles, q.M, q.colsum_a != nullptr);   // see gemm_tn_grouped_kernel
        int sp;
        if (isbig[i]) {
            sp = sp_big;
            const int maxs = q.K / 512;                              // >= 512 k rows per split
            if (sp > maxs) sp = maxs;
        } else {
            // share the block budget among the problems in proportion to their tiles; >= 64 rows per split
            int want = (int)((target_blocks * (long long)tiles / (tiles_total > 0 ? tiles_total : 1) + tiles - 1) / tiles);
            int maxs = cdiv(q.K > 0 ? q.K : 1, 64);
            sp = want < maxs ? want : maxs;
        }
        if (sp < 1) sp = 1;
        if (sp > 256) sp = 256;
        P.splits = sp;
        const int kq = (isbig[i] && q.K % 32 == 0) ? 32 : BKT;             // (32-deep stages of the big-tile kernel)
        P.kchunk = ((cdiv(q.K > 0 ? q.K : 1, sp) + kq - 1) / kq) * kq;
        P.tiles_n = cdiv(q.N, BN);
        const int fA = split4_flag(q.A, P.ra, q.M), fB = split4_flag(q.B, P.rb, q.N);    // XPS_FMT_SPLIT4 operands
        if (fA < 0 || fB < 0 || ((fA | fB) && !bf_mode())) return -2;
        P.vecA = (int)map_vec_ok(q.A, P.ra) | fA;
        P.vecB = (int)map_vec_ok(q.B, P.rb) | fB;
        P.slab_off = off;
        // LDS-DMA k loop: both operands XPS_FMT_SPLIT4 (the loop reads the split in place), whole 16-deep k-tiles per split
        // (kchunk and K multiples of 16: isbig), XPS_GEMM_DMA=0: the register-staged loop
        P.big = isbig[i] ? ((fA && fB && dma_enabled()) ? 2 : 1) : 0;
        if (isbig[i]) {
            P.block_start = blocks_big;
            blocks_big += (q.M / xps_big::TM) * (q.N / xps_big::TN) * sp;
        } else {
            P.block_start = blocks;
            blocks += tiles * sp;
        }
        off += (long long)sp * split_stride;
        out += split_stride;
    }
    g.total_blocks_big = blocks_big;
    g.total_blocks = blocks;
    g.total_out = out;
    ws_floats = (size_t)off;
    // same K everywhere: one common split count (the block budget over all tiles), blocks ordered by k-chunk first
    g.uniform = 0;
    g.tiles_all = 0;
    bool same_k = n > 1;
    for (int i = 1; i < n; ++i) same_k = same_k && probs[i].K == probs[0].K;
    // opt-in (XPS_TN_UNIFORM=1): measured twice (rounds 1 and 2, cfg-2 layer-1 group) -- same launch time (151 vs 152 us) and
    // the same fabric traffic within 1 %: co-resident blocks drift apart by more k-tiles than an XCD's L2 holds
    static const bool allow = [] { const char* e = getenv("XPS_TN_UNIFORM"); return e && e[0] == '1'; }();
    if (same_k && allow && probs[0].K > 0 && blocks_big == 0) {
        int sp = (int)((target_blocks + tiles_total - 1) / tiles_total);
        const int maxs = cdiv(probs[0].K, 64);
        if (sp > maxs) sp = maxs;
        if (sp < 1) sp = 1;
        if (sp > 256) sp = 256;
        const int kchunk = ((cdiv(probs[0].K, sp) + BKT - 1) / BKT) * BKT;
        blocks = 0; off = 0;
        int tstart = 0;
        for (int i = 0; i < n; ++i) {
            TnProb& P = g.p[i];
            const int tiles = cdiv(P.M, BM) * cdiv(P.N, BN);
            P.splits = sp; P.kchunk = kchunk;
            P.slab_off = off;
            P.block_start = blocks;          // (unused in uniform order)
            g.tile_start[i] = tstart;
            tstart += tiles;
            blocks += tiles * sp;
            off += (long long)sp * tn_split_stride(tiles, P.M, P.colsum != nullptr);
        }
        for (int i = n; i < TN_MAXP; ++i) g.tile_start[i] = 1 << 30;
        g.tiles_all = tstart;
        g.total_blocks = blocks;
        ws_floats = (size_t)off;
        g.uniform = 1;
    }
    return 0;
}
}  // namespace

extern "C" size_t xps_gemm_tn_grouped_f32_workspace(const xps_tn_problem* probs, int n) {
    // the larger of both tilings: the size must not depend on the precision mode / tile switch at launch time
    TnGroup g;
    size_t fl0 = 0, fl1 = 0;
    if (!probs || build_group(probs, n, g, fl0, 0) || build_group(probs, n, g, fl1, 1)) return 0;
    return (fl0 > fl1 ? fl0 : fl1) * sizeof(float) + 16;
}

extern "C" int xps_gemm_tn_grouped_f32(const xps_tn_problem* probs, int n, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    XPS_CHECK_ARG(probs, "null argument");
    TnGroup g;
    size_t fl = 0;
    if (build_group(probs, n, g, fl)) {
        xps_set_error("xps_gemm_tn_grouped_f32: invalid problem list (1..%d problems, non-null pointers, M,N >= 1; XPS_FMT_SPLIT4 "
                      "operands: bf16x3 mode, 16-byte aligned rows, M / N multiples of 4)", TN_MAXP);
        return XPS_E_INVALID;
    }
    if (!workspace || workspace_bytes < fl * sizeof(float) + 16 || !aligned16(workspace)) {
        xps_set_error("xps_gemm_tn_grouped_f32: workspace too small or misaligned");
        return XPS_E_WORKSPACE;
    }
    bool edge = false;
    for (int i = 0; i < n; ++i) edge = edge || (!g.p[i].big && ((probs[i].M % BM) || (probs[i].N % BN)));
    const bool bf = bf_mode();
    bool anypre = false;                   // a 128-tile problem with an XPS_FMT_SPLIT4 operand: the kernel instantiation that reads the flags
    for (int i = 0; i < n; ++i) anypre = anypre || (!g.p[i].big && ((g.p[i].vecA | g.p[i].vecB) & 2));
    if (g.total_blocks_big > 0) {
        static const bool ready = big_prepare(gemm_big_tn_kernel<false>, BIG_TN_LDS) && big_prepare(gemm_big_tn_kernel<true>, BIG_LDS32);
        if (!ready) {
            xps_set_error("xps_gemm_tn_grouped_f32: cannot reserve %d bytes of LDS", BIG_LDS32 > BIG_TN_LDS ? BIG_LDS32 : BIG_TN_LDS);
            return XPS_E_HIP;
        }
        // measured: the [k][x] x [k][x] form gains nothing from 32-deep stages (its fetches are full lines already) and loses
        // 1-5 % to the larger LDS block: 16-deep unless XPS_GEMM_BIG_DEEP_TN=1
        static const bool deep_tn = [] { const char* e = getenv("XPS_GEMM_BIG_DEEP_TN"); return e && e[0] == '1'; }();
        bool deep = big_deep_allowed() && deep_tn;
        for (int i = 0; i < n; ++i) deep = deep && (!g.p[i].big || (g.p[i].kchunk % 32 == 0 && g.p[i].K % 32 == 0 && !((g.p[i].vecA | g.p[i].vecB) & 2)));
        if (deep)
            hipLaunchKernelGGL(gemm_big_tn_kernel<true>, dim3(g.total_blocks_big), dim3(xps_big::NTHR), BIG_LDS32, (hipStream_t)stream, g,
                               (float*)workspace);
        else
            hipLaunchKernelGGL(gemm_big_tn_kernel<false>, dim3(g.total_blocks_big), dim3(xps_big::NTHR), BIG_TN_LDS, (hipStream_t)stream, g,
                           (float*)workspace);
        XPS_CHECK_LAUNCH();
    }
    if (g.total_blocks == 0) {
    } else if (bf && edge && anypre)
        hipLaunchKernelGGL((gemm_tn_grouped_kernel<true, true, true>), dim3(g.total_blocks), dim3(256), 0, (hipStream_t)stream, g, (float*)workspace);
    else if (bf && anypre)
        hipLaunchKernelGGL((gemm_tn_grouped_kernel<false, true, true>), dim3(g.total_blocks), dim3(256), 0, (hipStream_t)stream, g, (float*)workspace);
    else if (bf && edge)
        hipLaunchKernelGGL((gemm_tn_grouped_kernel<true, true>), dim3(g.total_blocks), dim3(256), 0, (hipStream_t)stream, g, (float*)workspace);
    else if (bf)
        hipLaunchKernelGGL((gemm_tn_grouped_kernel<false, true>), dim3(g.total_blocks), dim3(256), 0, (hipStream_t)stream, g, (float*)workspace);
    else if (edge)
        hipLaunchKernelGGL((gemm_tn_grouped_kernel<true, false>), dim3(g.total_blocks), dim3(256), 0, (hipStream_t)stream, g, (float*)workspace);
    else
        hipLaunchKernelGGL((gemm_tn_grouped_kernel<false, false>), dim3(g.total_blocks), dim3(256), 0, (hipStream_t)stream, g, (float*)workspace);
    XPS_CHECK_LAUNCH();
    int max_splits = 1;
    for (int i = 0; i < n; ++i) max_splits = g.p[i].splits > max_splits ? g.p[i].splits : max_splits;
    static const int red_mode = [] { const char* e = getenv("XPS_TN_REDUCE"); return !e ? 0 : (e[0] == 'f' ? 1 : (e[0] == 'l' ? 2 : 0)); }();
    if (red_mode == 1 || (red_mode == 0 && max_splits <= 32))
        hipLaunchKernelGGL(gemm_tn_grouped_reduce_flat, dim3(cdiv(g.total_out, 256 * 4)), dim3(256), 0, (hipStream_t)stream, g,
                           (const float*)workspace);
    else
        hipLaunchKernelGGL(gemm_tn_grouped_reduce, dim3(cdiv(g.total_out, RED_OUT * 4)), dim3(256), 0, (hipStream_t)stream, g,
                       (const float*)workspace);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

int xps_internal_gemm_mode() { return gemm_mode().load(std::memory_order_relaxed); }

extern "C" int xps_set_gemm_precision(int mode) {
    XPS_CHECK_ARG(mode == 0 || mode == 1, "mode: 0 = fp32 MFMA, 1 = bf16 split products");
    gemm_mode().store(mode);
    return XPS_OK;
}

extern "C" int xps_get_gemm_precision(void) { return gemm_mode().load(); }

extern "C" int xps_set_gemm_big_tiles(int on) {
    XPS_CHECK_ARG(on == 0 || on == 1, "on: 0 = 128 x 128 tiles only, 1 = 256 x 256 tiles for large interior shapes");
    big_switch().store(on);
    return XPS_OK;
}

extern "C" int xps_get_gemm_big_tiles(void) { return big_switch().load(); }

#ifdef XPS_GSTAMP
extern "C" int xps_debug_read_gstamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gstamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -2;
}
#endif
