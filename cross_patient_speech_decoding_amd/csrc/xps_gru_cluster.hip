// Cluster-persistent GRU recurrence for 256 < H <= 512 on gfx950 (the north-star shape: H = 512 / 500,
// nn_models/models.py:661-699).  W_hh (3H x H, 3 MB at H = 512) does not fit one CU, so a CLUSTER of
// CS = 16 workgroups (one per CU) shares it: member m keeps the rows of its 32 hidden units of all three gates
// resident in the VGPRs of its four CONTRACTION waves for the WHOLE sequence (192 registers per lane: fp32, or
// the bf16 hi/lo split made once per launch) and the cluster works on Mc trials.  Per step a member needs the
// complete previous state h_{t-1} of its trials (forward) or the complete gate gradients (backward): the
// members exchange them through a ping-pong buffer in global memory (L2), inside the launch.  DESIGN.md 4.1.
//
//   grid    : ndir x nblk clusters of 16 workgroups x 512 threads, <= one workgroup per CU
//   round   : 32 trials (two 16-trial MFMA column tiles); the operand image of a round (32 x 512 states, split
//             into bf16 hi / lo planes by the PRODUCER; fp32 mode: f32) is one contiguous 64 KiB, moved by
//             LDS-DMA (global_load_lds_dwordx4 sc1, 1-KiB pieces) into a double-buffered, bank-conflict-free
//             LDS image.
//   waves   : 0-3 contract (wave (ut, kh): 16 units x 3 gates x 256 k; 144 MFMAs per round: v_mfma_f32_16x16x32_bf16
//             on split operands lo*hi + hi*lo + hi*hi, fp32 mode 16x16x4 f32) and hand the products over through LDS;
//             4-7 sum the k-halves, run the gate math of 8 trials x the member's 32 units each, store outputs and exchange
//             planes, prefetch the gate inputs two rounds ahead, publish flags.
//             Gate-wave lane (t8, uq) owns trial 8 w + t8 of the round and FOUR CONSECUTIVE units 4 uq .. 4 uq + 3: every
//             global access is a 16-byte vector and a wave-instruction covers eight FULL 128-byte lines of its stream
//             (a 16-unit x 16-trial tile per wave touched sixteen half lines: BPTT launch 909 -> 808 us, forward 633 -> 604).
//
// Hand-off protocol (MI355X_MICROARCH.md "Valid forms", first table row; cdna_hip_programming.md G16 R1):
// exchanged bytes are loaded with sc1 (L1 bypass) and stored write-through (sc1) -- or with plain write-back
// stores when the cluster has verified at run time that all members sit on one XCD (one L2).  A member
// publishes round r of step s by ONE lane's sc1 flag store after every storing wave has waited for its
// exchange stores (counted s_waitcnt vmcnt: vmcnt retires in issue order and the issue order of a gate wave
// is pinned by CL_FENCE) and the workgroup barrier; consumers poll the 16 flags of a round with one sc1 load
// by one wave, the others load behind a barrier that wave joins afterwards.  Publication lags and polls lead
// two rounds, so in steady state nobody waits; the dependency chain needs NR >= 6 rounds per cluster (cl_plan).
// Every spin is bounded (status word, checked by the caller).  XPS_GRU_CLUSTER=steps runs the SAME kernels one
// step per launch (no in-kernel hand-off at all) and must give the same bits.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "xps_common.h"
#include <atomic>
#include "xps_gemm_tile.h"
using xps_tile::bf16x4;
using xps_tile::bf16x8;
using xps_tile::bf_split;

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int AUX_SC1 = 16;                 // cache-policy bits of the raw buffer builtins on gfx94x/gfx950: bit 4 = sc1
constexpr unsigned RSRC_FLAGS = 0x00020000; // buffer descriptor dword 3 for gfx9 raw buffers

template <int KSPLIT, bool BF, int KSEG>
struct ClCfg {
    static constexpr int KP = 256 * KSPLIT;                  // padded hidden size (k extent of one gate segment)
    static constexpr int NUT = 4 / KSPLIT;                   // 16-unit tiles per workgroup
    static constexpr int U = 16 * NUT;                       // hidden units per workgroup
    // One round's operand (32 trials x KP, bf16 hi plane + lo plane or f32: 4 bytes per element either way) is moved by
    // LDS-DMA in 1-KiB pieces (one wave instruction: 64 lanes x 16 B, contiguous in LDS).  Global order of a round:
    // [trial32][plane][KP] bf16 / [trial32][KP] f32.  LDS image: trials TS bytes apart with TS = 8 dwords (mod 64): the
    // b128 fragment reads of 16 trials x 4 k-quarters are bank-conflict free; pads sit between pieces.
    static constexpr int CHUNK_BYTES = 32 * KP * 4;          // one round in global memory
    static constexpr int NPIECE = CHUNK_BYTES / 1024;        // DMA pieces per round (64 or 32)
    static constexpr int TRIAL_BYTES = KP * 4;               // 2048 or 1024
    static constexpr int PPT = TRIAL_BYTES / 1024;           // pieces per trial (2 or 1)
    static constexpr int TS = TRIAL_BYTES + 32;              // trial stride in LDS (2080 or 1056)
    static constexpr int PS = BF ? (PPT == 2 ? 1040 : 512) : 0;            // plane stride (bf16): KP = 512: its own padded piece
    static constexpr int TILE_BYTES = 32 * TS;
    // products handed from the contraction waves to the gate-math waves: [wave 4][tile 2][gate 3] x 1 KiB (forward, one buffer +
    // a consumed word per gate-math wave) or [parity 2][wave 4][tile 2] x 1 KiB (backward)
    static constexpr int XACC_BYTES = KSEG == 1 ? 4 * 2 * 3 * 1024 : 2 * 4 * 2 * 1024;
    static constexpr int LDS_BYTES = 2 * TILE_BYTES + XACC_BYTES + 64;
    static constexpr int KSEGS = KSEG;                       // gate segments of the contraction (forward 1, backward 3)
    // LDS offset of DMA piece j inside a round image
    __host__ __device__ static constexpr int piece_off(int j) {
        return PPT == 2 ? (j >> 1) * TS + (j & 1) * (BF ? PS : 1024) : j * TS;
    }
};

#ifdef XPS_CL_STAMP
// Diagnostic build only (tools/stamp_cluster.py; never shipped): per-wave cycle sums of the loop segments, written to a
// buffer of their own that no kernel reads.
__device__ unsigned long long g_clstamp[2048 * 8];
__device__ unsigned long long g_clstamp2[2048 * 8];   // second set: phases inside the gate waves' slots (2-D BPTT kernel)
#define CL_STAMP(var)                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);
#define CL_ACC(sum, a, b) sum += (b) - (a);
#else
#define CL_STAMP(var)
#define CL_ACC(sum, a, b)
#endif

__device__ inline float cl_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ inline float cl_tanh(float x) { return 2.0f * cl_sigmoid(2.0f * x) - 1.0f; }

// one wave polls the CS flags of a round until every member has published `need`; bounded (3 s).  Slow path only (the
// look-ahead poll found the round unpublished): status[1] counts such waits, status[2] sums and status[3] keeps the longest
// of them in 10 ns ticks (diagnostics read by tools/bench_gru.py; status[0] != 0 means a wait gave up).
__device__ inline void cl_wait(const unsigned* flags, unsigned need, int cs, int lane, unsigned first, unsigned* status, unsigned* sticky) {
    unsigned v = first;
    if (__all(v >= need)) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    bool ok = false;
    for (;;) {
        v = need;
        if (lane < cs) v = __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__all(v >= need)) { ok = true; break; }
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) break;
    }
    if (lane == 0) {
        const unsigned dt = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0);
        if (!ok) {
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (sticky) __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __hip_atomic_fetch_add(status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(status + 2, dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_max(status + 3, dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Are all members of this cluster on ONE XCD?  Verified at run time, never assumed from blockIdx: every member publishes
// its XCC id, one wave collects the CS ids.  Same XCD: the members share an L2, so exchange stores may stay write-back
// (plain) and the sc1 loads (which bypass L1 only) hit that L2 instead of going to the memory side.  Otherwise every
// exchange store is write-through (sc1).  All members read the same table, so the whole cluster takes the same decision.
__device__ inline bool cl_same_xcd(unsigned* tab, int member, int cs, int lane, bool leader, unsigned* status, unsigned* sticky, unsigned* lds_word) {
    if (leader) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        id = (id & 15u) + 1u;
        if (lane == 0) __hip_atomic_store(tab + member, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned v = id;
        bool ok = false;
        for (;;) {
            v = id;
            if (lane < cs) v = __hip_atomic_load(tab + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v != 0u)) { ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) break;
        }
        if (!ok && lane == 0) {
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (sticky) __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool same = ok && __all(v == id);
        if (lane == 0) {
            *lds_word = same ? 1u : 0u;
            __hip_atomic_fetch_add(status + (same ? 4 : 5), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // diagnostics: workgroups on the one-XCD / mixed path
        }
    }
    __syncthreads();
    const bool r = *lds_word != 0u;
    __syncthreads();
    return r;
}

struct ClMap { int cluster, member; };
__device__ inline ClMap cl_map(int CS) {
    // members of a cluster share blockIdx % 8 (one XCD under round-robin placement) when the grid allows: speed only
    const int G = gridDim.x, bid = blockIdx.x;
    ClMap m;
    if (G % (8 * CS) == 0) {
        const int xcd = bid & 7, slot = bid >> 3;
        m.cluster = xcd * (G / 8 / CS) + slot / CS;
        m.member = slot % CS;
    } else {
        m.cluster = bid / CS;
        m.member = bid % CS;
    }
    return m;
}

// Four 1-KiB LDS-DMA pieces: each moves 64 lanes x 16 B from per-lane global addresses (gsrc, + 1024, + 2048, + 3072;
// L1 bypassed: sc1) to LDS [M0 + the same immediate, + 1024): the pads of the LDS image are folded into the M0 values.
// Inline asm: with the builtin the compiler orders EVERY later LDS read of the wave behind a piece with vmcnt(0).  Only the
// contraction waves issue pieces; they issue no other vector-memory operation in their loop and drain with vmcnt(0) before
// the round's barrier.  M0 is saved and restored (compiler-reserved).
__device__ inline void cl_dma4(const unsigned char* gsrc, unsigned l0, unsigned l1, unsigned l2, unsigned l3) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\t"
                 "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1024 sc1\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:2048 sc1\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:3072 sc1\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(l0), "s"(l1), "s"(l2), "s"(l3));
}

// ALTERNATIVE exchange-row layout of the forward kernel (bf16x3 mode, compile-time: -DXPS_CL_FWD_PLANES=0; round 4): XPS_FMT_SPLIT4
// groups (16 B: hi[0..3] | lo[0..3] of four consecutive k) instead of a hi plane and a lo plane, so that a gate-wave lane stores
// ONE 16-byte vector and a member's 32 units are one FULL 128-byte line of the row (the planes: two 8-byte stores, two half lines).
// An MFMA fragment needs the hi halves of TWO neighbouring groups: inside every 128-byte line (32 k = 8 groups) the row holds the
// four EVEN groups first, then the four odd ones -- the producer stores group g of a line at slot (g & 1) * 4 + (g >> 1)
// (cl_xslot), the DMA pieces copy lines as they lie; the k-quarter kq of a chunk reads groups 2 kq, 2 kq + 1 at line + 16 kq and
// + 64 (per instruction the bank pattern of a plane read: conflict-free) and regroups the halves in registers.
// Bit-identical, all cluster tests green -- and measured LEVEL on one box, three interleaved runs each (tools/bench_gru.py with
// XPS_LIB_OVERRIDE): forward with saved gates 591-597 us both ways, forward without 478 vs 464 us (slower), configs[3] step 6.51-6.52
// vs 6.49-6.53 ms.  The same layout in the BPTT kernel cost its launches 5-8 % (the contraction waves, which pace it, wait for
// both reads of a fragment and regroup them before the first MFMA), as did a per-lane group permutation on the source side of the
// DMA pieces.  The planes stay the default.
__device__ inline unsigned cl_xslot(unsigned g) { return ((g & 1u) << 2) | ((g & 7u) >> 1); }
#ifndef XPS_CL_BWD_LINEPLANAR
#define XPS_CL_BWD_LINEPLANAR 0     // exchange rows of the 1-D BPTT kernel (bf16x3): 0 = hi plane | lo plane, 1 = line-planar (see its epilogue; measured slower)
#endif
#ifndef XPS_CL_FWD_PLANES
#define XPS_CL_FWD_PLANES 1        // 0: the forward kernel's exchange rows as XPS_FMT_SPLIT4 groups (see above; A/B builds)
#endif

// image exchange (XIMG): lane 8 l + s of a DMA piece writes LDS slot s of line l and fetches global group 2 s (s < 4) or
// 2 (s - 4) + 1 of that line: the standard XPS_FMT_SPLIT4 row in memory becomes the even | odd arrangement in LDS
__device__ inline int cl_ximg_lane(int lane) { return (lane & ~7) | ((lane & 3) * 2 + ((lane >> 2) & 1)); }

// two trials' rows from two base addresses (the image's rows lie ndir * H * 4 bytes apart: beyond the instruction's offset field)
__device__ inline void cl_dma2x2(const unsigned char* g0, const unsigned char* g1, unsigned l0, unsigned l1, unsigned l2, unsigned l3) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1024 sc1\n\t"
                 "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off sc1\n\t"
                 "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off offset:1024 sc1\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g0), "v"(g1), "s"(l0), "s"(l1), "s"(l2), "s"(l3));
}

__device__ inline unsigned cl_lds_base(const unsigned char* smem) {
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) unsigned char*)smem;
}

// s_waitcnt vmcnt(n): all but this wave's n YOUNGEST vector-memory operations are done (they retire in issue order).  The
// gate-math waves issue per round, in this order (compiler fences between the groups): flag store / poll load, the
// LDS-DMA pieces of the next operand, the exchange stores, then the output stores and the loads for a later round.  Before the
// hand-off barrier only the first groups must be complete, so the wait leaves exactly the operations issued AFTER the
// exchange stores in flight.  n is the number of such operations the code issues unconditionally (stores of dead lanes are
// dropped by the buffer range check, not branched around); a smaller n is always safe, it only waits longer.
__device__ inline void cl_wait_vmcnt(int n) {
    switch (n) {
#define CL_VMCNT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        CL_VMCNT_CASE(1) CL_VMCNT_CASE(2) CL_VMCNT_CASE(3) CL_VMCNT_CASE(4) CL_VMCNT_CASE(5) CL_VMCNT_CASE(6) CL_VMCNT_CASE(7) CL_VMCNT_CASE(8)
        CL_VMCNT_CASE(9) CL_VMCNT_CASE(10) CL_VMCNT_CASE(11) CL_VMCNT_CASE(12) CL_VMCNT_CASE(13) CL_VMCNT_CASE(14) CL_VMCNT_CASE(15) CL_VMCNT_CASE(16) CL_VMCNT_CASE(17) CL_VMCNT_CASE(18) CL_VMCNT_CASE(19) CL_VMCNT_CASE(20)
#undef CL_VMCNT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
#define CL_FENCE() asm volatile("" ::: "memory")
constexpr unsigned CL_OOB = 0xFFFFFFF0u;       // buffer offset beyond every descriptor's range: the access is dropped

// resident weight fragments of one wave: rows = units j0 + n of `wsrc` rows (row stride ld floats), three k segments of 256
template <bool BF>
struct ClWeights {
    bf16x8 wh[BF ? 3 : 1][BF ? 8 : 1], wl[BF ? 3 : 1][BF ? 8 : 1];
    float wf[BF ? 1 : 3][BF ? 1 : 16][4];
    // seg_ptr(g): first element of segment g in this lane's row; kvalid: number of valid k from there (multiple of 4, <= 256)
    __device__ inline void load(const float* const (&seg)[3], bool rlive, int kq, int kvalid, const float* safe) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            if constexpr (BF) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int k = 32 * c + 8 * kq;
                    const bool ok0 = rlive && k + 3 < kvalid, ok1 = rlive && k + 7 < kvalid;
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(ok0 ? seg[g] + k : safe);
                    f32x4 v1 = *reinterpret_cast<const f32x4*>(ok1 ? seg[g] + k + 4 : safe);
                    if (!ok0) v0 = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (!ok1) v1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        __bf16 a, b;
                        bf_split(v0[e], a, b); wh[g][c][e] = a; wl[g][c][e] = b;
                        bf_split(v1[e], a, b); wh[g][c][4 + e] = a; wl[g][c][4 + e] = b;
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const int k = 16 * c + 4 * kq;
                    const bool ok = rlive && k + 3 < kvalid;
                    f32x4 v = *reinterpret_cast<const f32x4*>(ok ? seg[g] + k : safe);
                    if (!ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                    wf[g][c][0] = v[0]; wf[g][c][1] = v[1]; wf[g][c][2] = v[2]; wf[g][c][3] = v[3];
                }
            }
            // one segment's loads in flight at a time: the fp32 temporaries of all three would not fit beside the fragments
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};

// ------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------
struct ClFwd {
    const float* gi;
    const float* w_hh[2];
    const float* b_hh[2];
    float* y_ext;
    float* saved;
    void* xbuf;            // [2 parity][ndir][Bp] rows of KP elements (bf16 hi plane + lo plane, or f32)
    unsigned* flags;       // [cluster][NR][16]
    unsigned* xcc;         // [cluster][16]  XCC id + 1 of every member (cl_same_xcd)
    unsigned* status;
    unsigned* sticky;      // per-device word that outlives the workspace (xps_gru_set_status_word) or NULL
    unsigned xbuf_bytes;
    int T, B, H, ndir, Bp, Mc, NR, nblk, CS;
    int s_begin, s_end, handoff;
    int saved_mm;          // saved gates member-major (H % 32 == 0), see the epilogue
    // bf16x3 mode, optional (NULL: not written): XPS_FMT_SPLIT4 images for the GEMMs that read this layer's states -- of y_ext
    // (all T + 2 slots: h_prev of the dW_hh products) and of dropout(y) (T slots: the next layer's projection and dW_ih) --
    // written by the epilogue that holds the values, instead of one pass each over 168 MB (configs[3]) afterwards
    float* y_split;
    float* yd_split;
    float drop_p, drop_scale;
    unsigned long long drop_seed;
};

// 512 threads, two kinds of waves; every SIMD holds one of each, so the matrix pipe never waits for a memory latency or for
// the gate math:
//   waves 0-3  CONTRACT: resident W_hh fragments (unit tile ut = w / 2, k-half kh = w % 2), LDS operand reads, MFMA, products
//              to xacc, and between the MFMA chunks their quarter of the LDS-DMA pieces of the NEXT round's operand (the only
//              vector-memory operations they issue: drained with vmcnt(0) before the barrier);
//   waves 4-7  GATES: wave 4 + h owns (unit tile h / 2, trial tile h % 2): sum of the two k-halves from xacc, gate math,
//              exchange rows, outputs, requests for the gate inputs of later rounds, flags (wave 4).  All their vector-memory
//              operations are visible to the compiler (its waits for the input loads are exact) and only the exchange rows
//              must be complete at the barrier (cl_wait_vmcnt): output stores and input loads stay in flight across rounds.
// Round `it` = (step s, trials [32 r, 32 r + 32) of the cluster); one barrier per round.  During round it: contraction of
// round it (buffer it & 1), pieces of round it + 1 into the other buffer, gate math of round it - 1, flag of round it - 2
// (its exchange rows were complete before the last barrier), poll of the flags of round it + 2.
// XIMG (bf16x3 mode, H == KP, B % 32 == 0, training): the exchange buffer IS the XPS_FMT_SPLIT4 image of y_ext (ClFwd::y_split) -- a
// member publishes h_t by ONE 16-byte store per lane into slot t + 1 of the image (standard group order), the operand image of the
// next step is moved from the rows of that slot by LDS-DMA pieces whose lanes pick the groups in even / odd order (cl_ximg_lane; the
// fragment reads are those of the XPS_CL_FWD_PLANES = 0 layout), and the weight-gradient GEMMs read h_prev from the same image: no
// ring buffer traffic of its own, one store per lane and round fewer, and no xps_split4_f32 pass over y_ext afterwards.
template <int KSPLIT, bool BF, bool XIMG = false>
__global__ __launch_bounds__(512, 2) void gru_cluster_fwd_kernel(ClFwd p) {
    static_assert(KSPLIT == 2, "the cluster kernels cover 256 < H <= 512");
    static_assert(!XIMG || BF, "the image exchange exists in bf16x3 mode only");
    constexpr bool SPLITROWS = XIMG || !XPS_CL_FWD_PLANES;       // exchange rows / LDS image as split4 groups (even | odd per line)
    using Cf = ClCfg<KSPLIT, BF, 1>;
    constexpr int KP = Cf::KP, TS = Cf::TS, PS = Cf::PS, TILE = Cf::TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xacc = smem + 2 * TILE;
    unsigned* consumed = reinterpret_cast<unsigned*>(xacc + Cf::XACC_BYTES);      // [4] rounds taken out of xacc, per gate wave
    const unsigned lds0 = cl_lds_base(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const ClMap cm = cl_map(p.CS);
    const int dir = cm.cluster / p.nblk, blk = cm.cluster % p.nblk;
    const int T = p.T, B = p.B, H = p.H, NR = p.NR;
    const int ldy = p.ndir * H;
    const int m_base = blk * p.Mc;
    const int it_begin = p.s_begin * NR, it_end = p.s_end * NR;
    unsigned* myflags = p.flags + (long long)cm.cluster * NR * 16;
    if (tid < 4) consumed[tid] = 0u;

    // one step per launch: the kernel boundary publishes everything; persistent: write-back stores only inside one XCD
    const bool fast = !p.handoff || cl_same_xcd(p.xcc + cm.cluster * 16, cm.member, p.CS, lane, wave == 4, p.status, p.sticky, reinterpret_cast<unsigned*>(smem));

    // gate wave 4 + h moves pieces [16 h, 16 h + 16) of a round (trials 8 h .. 8 h + 7 of the 32) in four groups of four
    const int mvw = wave & 3;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.xbuf);
    const long long xrow = XIMG ? (long long)ldy * 4 : (long long)KP * 4;      // bytes between the rows of two trials
    auto dma_src = [&](int s_, int r_) -> const unsigned char* {
        if constexpr (XIMG) {
            // rows of slot_prev(s_) of the image: trial b = m_base + 32 r_ + 8 mvw + ..., this direction's H = KP elements
            const int slot_prev = (dir == 0) ? s_ : T + 1 - s_;
            return reinterpret_cast<const unsigned char*>(p.y_split) +
                   (((long long)slot_prev * B + m_base + 32 * r_ + 8 * mvw) * ldy + dir * H) * 4 + cl_ximg_lane(lane) * 16;
        }
        return xb + (size_t)(((s_ & 1) * p.ndir + dir) * p.Bp + m_base + 32 * r_) * (KP * 4) + (size_t)(mvw * 16) * 1024 + lane * 16;
    };
    auto dma_group = [&](const unsigned char* src, int buf, int grp) {      // grp = 0 .. 3 (compile time at every call site)
        const int j = mvw * 16 + grp * 4;                                   // first piece: a multiple of four = trial boundary
        const unsigned base = lds0 + (unsigned)(buf * TILE) + (unsigned)((j >> 1) * TS);
        // pieces j .. j + 3 = (trial, half) (t, 0), (t, 1), (t + 1, 0), (t + 1, 1); the immediates add 0, 1024, 2048, 3072
        const unsigned h = BF ? PS : 1024;
        if constexpr (XIMG) cl_dma2x2(src + (2 * grp) * xrow, src + (2 * grp + 1) * xrow, base, base + h - 1024, base + TS, base + TS + h - 1024);
        else cl_dma4(src + grp * 4096, base, base + h - 1024, base + TS - 2048, base + TS + h - 3072);
    };

    if (wave < 4) {
        const unsigned char* src = dma_src(p.s_begin, 0);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) dma_group(src, it_begin & 1, g4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#ifdef XPS_CL_STAMP
    unsigned long long sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0, s_bar = 0, s_work = 0, s_drain = 0;
    CL_STAMP(sb2)
#define CL_STORE_STAMPS() if (lane == 0) { const int wid = (blockIdx.x * 8 + wave) & 2047; g_clstamp[wid * 8 + 0] = s_work; g_clstamp[wid * 8 + 1] = s_drain; g_clstamp[wid * 8 + 2] = s_bar; }
#else
#define CL_STORE_STAMPS()
#endif

    if (wave < 4) {
        // ---------------- contraction waves ----------------
        const int ut = wave >> 1, kh = wave & 1;
        const int j0 = cm.member * Cf::U + ut * 16;
        const int kbase = kh * 256;
        const float* __restrict__ W = p.w_hh[dir];
        ClWeights<BF> w;
        {
            const int jr = j0 + n;
            const bool rlive = jr < H;
            const int jrc = rlive ? jr : 0;
            const float* const seg[3] = {W + (long long)(0 * H + jrc) * H + kbase, W + (long long)(1 * H + jrc) * H + kbase,
                                         W + (long long)(2 * H + jrc) * H + kbase};
            w.load(seg, rlive, kq, H - kbase, W);
        }
        CL_STAMP(sb2)
        int s_nx = p.s_begin, r_nx = 0;                 // (step, round) of round it + 1
        for (int it = it_begin; it < it_end; ++it) {
            const unsigned char* tb = smem + (it & 1) * TILE;
            // the operand image of round it + 1, first thing (its flags were polled by wave 4 two rounds ago): this wave's 16 pieces;
            // the MFMAs below cover their landing, the counted wait before the barrier finds them complete
            if (++r_nx == NR) { r_nx = 0; ++s_nx; }
            if (it + 1 < it_end) {
                const unsigned char* src = dma_src(s_nx, r_nx);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) dma_group(src, (it + 1) & 1, g4);
            }
            CL_FENCE();
            f32x4 acc[2][3];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[tt][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) {
                // 16 steps (tile, chunk); the fragments of step i + 2 are requested before the MFMAs of step i (a ring of three,
                // pinned with scheduling barriers: left alone the scheduler issues the reads of two steps right in front of
                // their 18 MFMAs and exposes the LDS latency eight times per round)
                // (exchange rows are XPS_FMT_SPLIT4 groups; a line holds its even groups, then its odd ones: cl_xslot)
                const unsigned char* rp0 = !SPLITROWS ? tb + n * TS + (kbase + 8 * kq) * 2 : tb + n * TS + kh * PS + kq * 16;
                bf16x8 bh[3], bl[3];
                auto frag = [&](int i, bf16x8& h8, bf16x8& l8) {
                    if constexpr (!SPLITROWS) {
                        const unsigned char* rp = rp0 + (i >> 3) * 16 * TS + (i & 7) * 64;
                        h8 = *reinterpret_cast<const bf16x8*>(rp);
                        l8 = *reinterpret_cast<const bf16x8*>(rp + PS);
                        return;
                    }
                    const unsigned char* rp = rp0 + (i >> 3) * 16 * TS + (i & 7) * 128;
                    const bf16x8 ga = *reinterpret_cast<const bf16x8*>(rp);
                    const bf16x8 gb = *reinterpret_cast<const bf16x8*>(rp + 64);
                    h8 = __builtin_shufflevector(ga, gb, 0, 1, 2, 3, 8, 9, 10, 11);
                    l8 = __builtin_shufflevector(ga, gb, 4, 5, 6, 7, 12, 13, 14, 15);
                };
                frag(0, bh[0], bl[0]);
                frag(1, bh[1], bl[1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int tt = i >> 3, c = i & 7;
                    if (i + 2 < 16) frag(i + 2, bh[(i + 2) % 3], bl[(i + 2) % 3]);
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8 h8 = bh[i % 3], l8 = bl[i % 3];
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        acc[tt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.wl[g][c], h8, acc[tt][g], 0, 0, 0);
                        acc[tt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.wh[g][c], l8, acc[tt][g], 0, 0, 0);
                        acc[tt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.wh[g][c], h8, acc[tt][g], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(tb + (tt * 16 + n) * TS + (kbase + 16 * c + 4 * kq) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int g = 0; g < 3; ++g)
                                acc[tt][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.wf[g][c][e], a4[e], acc[tt][g], 0, 0, 0);
                    }
                }
            }
            // the previous round's products must have been taken out of xacc (they were, two thousand cycles ago: one look)
            if (it > it_begin) {
                const unsigned want = (unsigned)(it - it_begin);
                // (every gate wave takes units of both unit tiles)
                while (__hip_atomic_load(consumed + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want ||
                       __hip_atomic_load(consumed + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want ||
                       __hip_atomic_load(consumed + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want ||
                       __hip_atomic_load(consumed + 3, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want)
                    __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    *reinterpret_cast<f32x4*>(xacc + ((wave * 2 + tt) * 3 + g) * 1024 + lane * 16) = acc[tt][g];
            CL_STAMP(sb3)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the next image have landed
            CL_STAMP(sb0)
            __syncthreads();
            CL_STAMP(sb1)
            CL_ACC(s_bar, sb0, sb1) CL_ACC(s_work, sb2, sb3) CL_ACC(s_drain, sb3, sb0)
#ifdef XPS_CL_STAMP
            sb2 = sb1;
#endif
        }
        CL_STORE_STAMPS()
        return;
    }

    // ---------------- gate waves ----------------
    // They share their SIMD with a contraction wave that always has an MFMA ready; at equal priority the older contraction
    // wave wins every arbitration and the gate math (a few hundred instructions per round) takes longer than the round's
    // MFMAs.  With priority the gate wave's instructions go first whenever it has one, the MFMAs fill the rest.
    __builtin_amdgcn_s_setprio(3);
    const int hw = wave - 4;
    // lane (t8, uq): trial 8 hw + t8 of the round's 32, units 4 uq .. 4 uq + 3 of the member's 32 (see the BPTT kernel)
    const int gtr = 8 * hw + (lane >> 3);
    const int guq = lane & 7;
    const int ju = cm.member * Cf::U + 4 * guq;
    const bool ulive = ju < H;
    const int juc = ulive ? ju : 0;
    const float* __restrict__ gi = p.gi + (long long)dir * T * B * 3 * H;
    const bool has_saved = p.saved != nullptr;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.xbuf, 0, p.xbuf_bytes, RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y_ext, 0, (unsigned)((long long)(T + 2) * B * ldy * 4), RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc(p.saved, 0, has_saved ? (unsigned)((long long)p.ndir * T * B * 4 * H * 4) : 0u, RSRC_FLAGS);
    // (absent images: empty descriptors, the stores below are issued and dropped like those of dead lanes)
    __amdgpu_buffer_rsrc_t ysr = __builtin_amdgcn_make_buffer_rsrc(p.y_split ? p.y_split : p.y_ext, 0, p.y_split ? (unsigned)((long long)(T + 2) * B * ldy * 4) : 0u, RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t ydr = __builtin_amdgcn_make_buffer_rsrc(p.yd_split ? p.yd_split : p.y_ext, 0, p.yd_split ? (unsigned)((long long)T * B * ldy * 4) : 0u, RSRC_FLAGS);
    const bool images = BF && !XIMG && (p.y_split || p.yd_split);       // extra image stores of the ring-buffer form (opt-in)
    const bool img_drop = XIMG && p.yd_split;                            // XIMG: the dropped image stays an extra store
    f32x4 bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        bias[g] = *reinterpret_cast<const f32x4*>(p.b_hh[dir] + g * H + juc);
        if (!ulive) bias[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    struct EpiIn { f32x4 gr, gz, gn; u32x4 hp; };
    auto epi_load = [&](int sn, int rn, EpiIn& in) {      // 4 loads, always issued
        const int t = (dir == 0) ? sn : T - 1 - sn;
        const int slot_prev = (dir == 0) ? t : t + 2;
        const int b = m_base + 32 * rn + gtr;
        const int bc = b < B ? b : B - 1;
        const float* gp = gi + ((long long)t * B + bc) * 3 * H + juc;
#ifdef XPS_CL_ABL_NOLOAD
        in.gr = (f32x4){0.1f, 0.2f, 0.3f, 0.4f}; in.gz = in.gr; in.gn = in.gr; in.hp = (u32x4){0u, 0u, 0u, 0u};
        (void)gp; (void)slot_prev;
#else
        in.gr = *reinterpret_cast<const f32x4*>(gp);
        in.gz = *reinterpret_cast<const f32x4*>(gp + H);
        in.gn = *reinterpret_cast<const f32x4*>(gp + 2 * H);
        // own previous state (fp32), written by this lane one step ago (or by the init kernel)
        in.hp = __builtin_amdgcn_raw_buffer_load_b128(yr, (unsigned)((((long long)slot_prev * B + bc) * ldy + dir * H + juc) * 4), 0, AUX_SC1);
#endif
    };
    // gates + hidden update of round (sn, rn) for this lane's trial and four units; exchange rows of the next step (every lane
    // stores: rows of pad trials / pad units carry zeros), then the outputs: h_t and (training) the saved gates, 1 or 5 stores,
    // ALWAYS issued (dead lanes: offset out of range, dropped)
    auto epilogue = [&](int sn, int rn, const EpiIn& in, const f32x4 (&a)[3]) {
        const int t = (dir == 0) ? sn : T - 1 - sn;
        const int b = m_base + 32 * rn + gtr;
        const bool live = b < B && ulive;
        const f32x4 hp = __builtin_bit_cast(f32x4, in.hp);      // (whole vector: a bit_cast of ONE element reads element 0)
        f32x4 o, rg, zg, ng, qv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rg[i] = cl_sigmoid(in.gr[i] + a[0][i] + bias[0][i]);
            zg[i] = cl_sigmoid(in.gz[i] + a[1][i] + bias[1][i]);
            qv[i] = a[2][i] + bias[2][i];
            ng[i] = cl_tanh(in.gn[i] + rg[i] * qv[i]);
            o[i] = live ? ng[i] + zg[i] * (hp[i] - ng[i]) : 0.f;
        }
        CL_FENCE();
        if constexpr (XIMG) {
            // the exchange row IS the image row of slot t + 1 (an output: written for the last step too); same offset as y_ext's
            const u32x4 sp = __builtin_bit_cast(u32x4, split4_pack(o));
            const unsigned xo = live ? (unsigned)((((long long)(t + 1) * B + b) * ldy + dir * H + ju) * 4) : CL_OOB;
            // (the same 16 bytes as two 8-byte stores: level, 619-629 vs 621-631 us per launch on one box)
            if (fast) __builtin_amdgcn_raw_buffer_store_b128(sp, ysr, xo, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(sp, ysr, xo, 0, AUX_SC1);
        } else if (sn + 1 < T) {
            const unsigned row = (unsigned)((((sn + 1) & 1) * p.ndir + dir) * p.Bp + b);
            if constexpr (BF) {
                // (SPLITROWS: ONE 16-byte store per lane, hi[0..3] | lo[0..3] of its four units; a member's 32 units = one full line)
                if constexpr (!SPLITROWS) {
                    bf16x4 sh, sl;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { __bf16 x, c; bf_split(o[i], x, c); sh[i] = x; sl[i] = c; }
                    const unsigned off = row * (unsigned)(KP * 4) + (unsigned)ju * 2u;
                    if (fast) {
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sh), xr, off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sl), xr, off + KP * 2, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sh), xr, off, 0, AUX_SC1);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sl), xr, off + KP * 2, 0, AUX_SC1);
                    }
                } else {
                    const u32x4 sp = __builtin_bit_cast(u32x4, split4_pack(o));
                    const unsigned off = row * (unsigned)(KP * 4) + ((unsigned)ju & ~31u) * 4u + cl_xslot((unsigned)ju >> 2) * 16u;
                    if (fast) __builtin_amdgcn_raw_buffer_store_b128(sp, xr, off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(sp, xr, off, 0, AUX_SC1);
                }
            } else {
                const unsigned off = row * (unsigned)(KP * 4) + (unsigned)ju * 4u;
                if (fast) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), xr, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), xr, off, 0, AUX_SC1);
            }
        }
        CL_FENCE();
        const unsigned yo = live ? (unsigned)((((long long)(t + 1) * B + b) * ldy + dir * H + ju) * 4) : CL_OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yr, yo, 0, 0);
        if constexpr (BF) {
            if (images || img_drop) {   // (wave-uniform; two stores / one, counted by the caller)
                if (!XIMG) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, split4_pack(o)), ysr, yo, 0, 0);
                const long long e = ((long long)t * B + b) * ldy + dir * H + ju;        // element index in (T, B, ndir * H)
                f32x4 v = o;
                // (same decisions and arithmetic as split4_kernel / dropout_kernel: xps_common.h dropout_keep4 on the quad index)
                if (p.drop_p > 0.f) v = v * dropout_keep4(p.drop_seed, e >> 2, p.drop_p) * p.drop_scale;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, split4_pack(v)), ydr, live ? (unsigned)(e * 4) : CL_OOB, 0, 0);
            }
        }
        if (has_saved) {
            // saved gates: private between this kernel and the BPTT kernels.  H % 32 == 0 (saved_mm): MEMBER-major --
            // [dir][t][member][trial][gate 4][32 units] -- so that a member streams 512 contiguous bytes per trial and its trials
            // back to back (row-major [trial][gate][H] gives it four 128-byte pieces 2 KB apart per trial: measured 8 % of the BPTT
            // launch in DRAM page misses); otherwise [dir][t][trial][4H]
            const unsigned gstride = p.saved_mm ? 128u : (unsigned)H * 4u;
            const unsigned so = !live ? CL_OOB : p.saved_mm
                ? (unsigned)((((((long long)dir * T + t) * p.CS + cm.member) * B + b) * 128 + (ju & 31)) * 4)
                : (unsigned)(((((long long)dir * T + t) * B + b) * 4 * H + ju) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rg), sr, so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, zg), sr, live ? so + gstride : CL_OOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ng), sr, live ? so + 2u * gstride : CL_OOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, qv), sr, live ? so + 3u * gstride : CL_OOB, 0, 0);
        }
    };
    // products of the finished round for this wave's trial tile: k-low half + k-high half, taken out of xacc
    f32x4 prod[3];
    auto take_products = [&](int rounds_done) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int ut_ = guq >> 2, te_ = gtr >> 4, lo_ = ((guq & 3) * 16 + (gtr & 15)) * 16;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(xacc + (((2 * ut_) * 2 + te_) * 3 + g) * 1024 + lo_);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(xacc + (((2 * ut_ + 1) * 2 + te_) * 3 + g) * 1024 + lo_);
            prod[g] = lo + hi;
        }
        // the reads above must have returned before the contraction waves may overwrite xacc
        if (lane == 0) __hip_atomic_store(consumed + hw, (unsigned)rounds_done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    // gate inputs live in three register slots (round index mod 3 relative to the launch's first round): requested during
    // round it - 2, used during round it + 1 (HBM latency under load exceeds one round).  The loop body is instantiated per
    // slot so that no copy (which would wait for the loads) and no dynamic register index is needed.  (step, round) of
    // rounds it - 1 and it + 2 are carried along.
    EpiIn ein[3];
    int s_pv = p.s_begin, r_pv = -1;                    // round it - 1
    int s_n2 = p.s_begin, r_n2 = 0;                     // round it + 2 (starts as round it_begin, advanced below)
    epi_load(s_n2, r_n2, ein[0]);
    if (++r_n2 == NR) { r_n2 = 0; ++s_n2; }
    if (it_begin + 1 < it_end) epi_load(s_n2, r_n2, ein[1]);
    if (++r_n2 == NR) { r_n2 = 0; ++s_n2; }
    auto round_body = [&](int it, auto SLOT) {
        constexpr int cur = decltype(SLOT)::value;            // slot of round it; round it - 1: (cur + 2) % 3; round it + 2: the same slot
        constexpr int prv = (cur + 2) % 3;
        // oldest group: flag of round it - 2 (its exchange rows were complete before the last barrier), flags of round it + 2
        if (p.handoff && wave == 4 && lane == 0 && it >= it_begin + 2) {
            int rp = r_pv - 1, sp = s_pv;
            if (rp < 0) { rp = NR - 1; --sp; }
            __hip_atomic_store(myflags + rp * 16 + cm.member, (unsigned)(sp + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool do_poll = p.handoff && wave == 4 && it + 2 < it_end && s_n2 > p.s_begin;
        unsigned fl = 0xffffffffu;
        if (do_poll && lane < p.CS) fl = __hip_atomic_load(myflags + r_n2 * 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CL_FENCE();
        // (the LDS-DMA of round it + 1 is issued by the contraction waves, first thing in their round: here the pieces sat in
        //  front of the gate math, whose compiler-placed waits cannot see them and waited for their landing every round)
        int younger = 0;                                    // operations issued after the exchange stores (see cl_wait_vmcnt)
        if (it > it_begin) {
            take_products(it - it_begin);
            epilogue(s_pv, r_pv, ein[prv], prod);
            younger += (has_saved ? 5 : 1) + (images ? 2 : 0) + (img_drop ? 1 : 0);
        }
        CL_FENCE();
        if (it + 2 < it_end) {
            epi_load(s_n2, r_n2, ein[prv]);                 // (the slot the gate math above has just released)
#ifndef XPS_CL_ABL_NOLOAD
            younger += 4;
#endif
        }
        CL_FENCE();
        CL_STAMP(sb3)
        cl_wait_vmcnt(younger);                             // exchange rows complete; outputs / later inputs stay in flight
        if (do_poll) cl_wait(myflags + r_n2 * 16, (unsigned)s_n2, p.CS, lane, fl, p.status, p.sticky);
        CL_STAMP(sb0)
        __syncthreads();
        CL_STAMP(sb1)
        CL_ACC(s_bar, sb0, sb1) CL_ACC(s_work, sb2, sb3) CL_ACC(s_drain, sb3, sb0)
#ifdef XPS_CL_STAMP
        sb2 = sb1;
#endif
        if (++r_pv == NR) { r_pv = 0; ++s_pv; }
        if (++r_n2 == NR) { r_n2 = 0; ++s_n2; }
    };
    for (int it = it_begin; it < it_end; it += 3) {
        round_body(it, std::integral_constant<int, 0>{});
        if (it + 1 < it_end) round_body(it + 1, std::integral_constant<int, 1>{});
        if (it + 2 < it_end) round_body(it + 2, std::integral_constant<int, 2>{});
    }
    if (it_end > it_begin) {                                    // (outputs of the launch's last round; nobody waits for its flag)
        take_products(it_end - it_begin);
        const int slot = (it_end - 1 - it_begin) % 3;
        if (slot == 0) epilogue(s_pv, r_pv, ein[0], prod);
        else if (slot == 1) epilogue(s_pv, r_pv, ein[1], prod);
        else epilogue(s_pv, r_pv, ein[2], prod);
    }
    CL_STORE_STAMPS()
}

// h0 -> slots of y_ext and parity 0 of the exchange buffer (all Bp rows, all KP columns: pads are zero)
template <bool BF>
__global__ void gru_cluster_init_kernel(const float* __restrict__ h0, float* __restrict__ y_ext, void* __restrict__ xbuf,
                                        int T, int B, int H, int ndir, int Bp, int KP, u32x4* __restrict__ header, int header_u4,
                                        float* __restrict__ y_split) {
    // the workspace header (flags, XCC table, status block: header_u4 16-byte words) is zeroed here instead of by a memset of its
    // own in front of every launch (5 us each, five recurrence launches per configs[3] step)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < header_u4; i += gridDim.x * blockDim.x) header[i] = (u32x4){0u, 0u, 0u, 0u};
    const long long total = (long long)ndir * Bp * (KP / 4);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % (KP / 4)) * 4;
        const long long rb = i / (KP / 4);
        const int b = (int)(rb % Bp), dir = (int)(rb / Bp);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const bool in = b < B && k < H;
        if (in && h0) v = *reinterpret_cast<const f32x4*>(h0 + ((long long)dir * B + b) * H + k);
        if (in) {
            const int slot_h0 = (dir == 0) ? 0 : T + 1, slot_other = (dir == 0) ? T + 1 : 0;
            const int ldy = ndir * H;
            *reinterpret_cast<f32x4*>(y_ext + ((long long)slot_h0 * B + b) * ldy + dir * H + k) = v;
            *reinterpret_cast<f32x4*>(y_ext + ((long long)slot_other * B + b) * ldy + dir * H + k) = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (BF) {
                if (y_split) {          // the same two slots of the XPS_FMT_SPLIT4 image of y_ext (ClFwd::y_split)
                    *reinterpret_cast<f32x4*>(y_split + ((long long)slot_h0 * B + b) * ldy + dir * H + k) = split4_pack(v);
                    *reinterpret_cast<f32x4*>(y_split + ((long long)slot_other * B + b) * ldy + dir * H + k) = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (!xbuf) continue;                               // (image exchange: slot_h0 of y_split, written above, is the first operand)
        unsigned char* row = reinterpret_cast<unsigned char*>(xbuf) + ((long long)dir * Bp + b) * KP * 4;
        if constexpr (BF) {
            if (XPS_CL_FWD_PLANES) {
                bf16x4 sh, sl;
#pragma unroll
                for (int e = 0; e < 4; ++e) { __bf16 a, c; bf_split(v[e], a, c); sh[e] = a; sl[e] = c; }
                *reinterpret_cast<bf16x4*>(row + k * 2) = sh;
                *reinterpret_cast<bf16x4*>(row + KP * 2 + k * 2) = sl;
            } else {
                *reinterpret_cast<f32x4*>(row + (k & ~31) * 4 + cl_xslot((unsigned)k >> 2) * 16) = split4_pack(v);
            }
        } else {
            *reinterpret_cast<f32x4*>(row + k * 4) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward (BPTT): dh_{t-1} = z * dh_t + [dar | daz | dan*r] W_hh.  Member m owns U output units (rows of
// W_hh^T: 16 units x 3 x 256 k per wave) and the gate gradients of the same units; the contraction runs
// over all 3H gate-gradient columns, exchanged per step as three gate segments of KP columns.
// Processing step ps = 0 .. T-1 handles s = T-1-ps (t = s forward, T-1-s reverse); ps = T is the dh0 pass.
// Sub-iteration q = (ps, round r, gate segment g) contracts one 64-KiB operand image; the gate gradients of a round
// are computed one sub-iteration after its last segment (same schedule as the forward kernel's rounds).
// ------------------------------------------------------------------------------------------------------
struct ClBwd {
    const float* dy;
    const float* dhn;
    const float* y_ext;
    const float* saved;
    const float* w_hh_t[2];     // (H x 3H)
    float* dgi;
    float* dghn;
    float* dh0;
    float* keep;                // [ndir][B][H]  z * dh of the step processed before
    void* xbuf;                 // [2 parity][ndir][Bp/32][3 gates][32 trials] rows of KP elements (bf16 hi plane | lo plane, or f32)
    unsigned* flags;
    unsigned* xcc;
    unsigned* status;
    unsigned* sticky;
    unsigned xbuf_bytes;
    int T, B, H, ndir, Bp, Mc, NR, nblk, CS;
    int ps_begin, ps_end, ps_total, handoff;
    int saved_mm;               // saved gates member-major (see the forward kernel's epilogue)
    int split_out;              // != 0 (bf16x3 mode): dgi / dghn are written as XPS_FMT_SPLIT4 groups (xps.h) for the GEMMs that read them
};

// Same two wave roles as the forward kernel.  Sub-iteration q = (ps, round r, gate segment g): contraction waves accumulate
// segment g of round r (buffer q & 1), stream three quarters of their share of sub-iteration q + 1's operand between the MFMA
// chunks (bf16x3 mode; the gate waves move the fourth quarter, first thing in every sub-iteration; fp32 mode: all of it) and hand
// the finished products over after g == 2 (xacc, double buffered by round parity); gate waves act once per round, during its
// first sub-iteration: gate math of the PREVIOUS round (products from xacc, inputs requested one round earlier), exchange rows,
// outputs, then the requests for THIS round's inputs; wave 4 also keeps the flags.
// XOUT (bf16x3 mode, split4 outputs, H == KP, no pad trials): the OUTPUTS are the exchange rows -- dgi's r and z columns and dghn of
// step ps - 1 are exactly the gate gradients step ps contracts, and they are written as XPS_FMT_SPLIT4 groups anyway.  No exchange
// ring, no exchange stores (six 8-byte stores per lane and round less, 0.5 GB of write-through traffic per launch less); the
// operand images are moved from the rows of dgi / dghn (lanes of a piece in even / odd group order: cl_ximg_lane) and the
// contraction waves regroup the halves of two groups per fragment, as the forward kernel's XIMG form does.
template <int KSPLIT, bool BF, bool XOUT = false>
__global__ __launch_bounds__(512, 2) void gru_cluster_bwd_kernel(ClBwd p) {
    static_assert(KSPLIT == 2, "the cluster kernels cover 256 < H <= 512");
    static_assert(!XOUT || BF, "outputs as exchange rows: bf16x3 mode only");
    using Cf = ClCfg<KSPLIT, BF, 3>;
    constexpr int KP = Cf::KP, TS = Cf::TS, PS = Cf::PS, TILE = Cf::TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* xacc = smem + 2 * TILE;              // [round parity 2][contraction wave 4][tile 2] x 1 KiB
    const unsigned lds0 = cl_lds_base(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const ClMap cm = cl_map(p.CS);
    const int dir = cm.cluster / p.nblk, blk = cm.cluster % p.nblk;
    const int T = p.T, B = p.B, H = p.H, NR = p.NR;
    const int ldy = p.ndir * H;
    const int m_base = blk * p.Mc;
    const int NQ = 3 * NR;                              // sub-iterations (round, gate segment) per processing step
    unsigned* myflags = p.flags + (long long)cm.cluster * NR * 16;
    const bool fast = !p.handoff || cl_same_xcd(p.xcc + cm.cluster * 16, cm.member, p.CS, lane, wave == 4, p.status, p.sticky, reinterpret_cast<unsigned*>(smem));

    // gate waves: unit tile, trial tile, units, buffers
    const int hw = wave >= 4 ? wave - 4 : 0;
    // gate-wave lane (t8, uq): trial 8 hw + t8 of the round's 32, units 4 uq .. 4 uq + 3 of the member's 32: a wave-instruction
    // covers 8 trials x 32 units = eight FULL 128-byte lines of every stream (a 16-unit tile per wave touched sixteen half lines)
    const int gtr = 8 * hw + (lane >> 3);
    const int guq = lane & 7;
    const int ju = cm.member * Cf::U + 4 * guq;
    const bool ulive = ju < H;
    const int juc = ulive ? ju : 0;
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.xbuf, 0, p.xbuf_bytes, RSRC_FLAGS);
    // running gradient z * dh: private to this kernel, member-major [dir][member][trial][32 units] (contiguous per member)
    __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(p.keep, 0, (unsigned)((long long)p.ndir * p.CS * B * 32 * 4), RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(p.dgi, 0, (unsigned)((long long)p.ndir * T * B * 3 * H * 4), RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t nr = __builtin_amdgcn_make_buffer_rsrc(p.dghn, 0, (unsigned)((long long)p.ndir * T * B * H * 4), RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(p.dh0, 0, p.dh0 ? (unsigned)((long long)p.ndir * B * H * 4) : 0u, RSRC_FLAGS);
    const bool has_dy = p.dy != nullptr;

    // sub-iteration (ps, r, g) reads the gate gradients that processing step ps - 1 wrote (parity (ps - 1) & 1); contraction
    // wave w moves pieces [16 w, 16 w + 16) in four groups of four
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.xbuf);
    // (w: the contraction wave whose quarter of the pieces is meant; gate wave 4 + w moves the last group of wave w's quarter)
    const int dw = wave & 3;
    // XOUT: rows of the step processed before (time index t_prev) in dgi (segments r, z: 3H elements per trial) / dghn (segment n)
    // -- the trial stride (bytes) is folded into bit 0..: dgi rows lie 6 KB apart, dghn rows 2 KB; it travels in the low bits of
    // nothing: dma_group recomputes it from the segment it is given
    auto dma_src = [&](int ps_, int r_, int g_) -> const unsigned char* {
        if constexpr (XOUT) {
            const int sp = T - 1 - (ps_ - 1);
            const int tp = (dir == 0) ? sp : T - 1 - sp;
            const long long row = ((long long)dir * T + tp) * B + m_base + 32 * r_ + 8 * dw;
            const float* base = g_ < 2 ? p.dgi + row * 3 * H + g_ * H : p.dghn + row * H;
            return reinterpret_cast<const unsigned char*>(base) + cl_ximg_lane(lane) * 16;
        }
        const size_t chunk = (size_t)(((((ps_ - 1) & 1) * p.ndir + dir) * (p.Bp / 32) + (m_base / 32) + r_) * 3 + g_);
        return xb + chunk * Cf::CHUNK_BYTES + (size_t)(dw * 16) * 1024 + lane * 16;
    };
    auto dma_group = [&](const unsigned char* src, int buf, int grp, int g_ = 0) {
        const int j = dw * 16 + grp * 4;
        const unsigned base = lds0 + (unsigned)(buf * TILE) + (unsigned)((j >> 1) * TS);
        const unsigned h = BF ? PS : 1024;
        if constexpr (XOUT) {
            const long long st = (g_ < 2 ? 3LL : 1LL) * H * 4;           // bytes between the rows of two trials
            cl_dma2x2(src + (2 * grp) * st, src + (2 * grp + 1) * st, base, base + h - 1024, base + TS, base + TS + h - 1024);
        } else {
            cl_dma4(src + grp * 4096, base, base + h - 1024, base + TS - 2048, base + TS + h - 3072);
        }
    };

    struct EpiIn { f32x4 dy, rg, zg, ng, q, hp; u32x4 keep; };
    // inputs of the gate math of (ps, r): returns the number of loads issued (always the same for a given ps)
    auto epi_load = [&](int ps, int r, EpiIn& in) -> int {
        const int s = T - 1 - ps;
        const int t = (dir == 0) ? s : T - 1 - s;
        const int slot_prev = (dir == 0) ? t : t + 2;
        const int b = m_base + 32 * r + gtr;
        const int bc = b < B ? b : B - 1;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        int nload = 0;
        in.keep = (u32x4){0u, 0u, 0u, 0u};
        if (ps > 0) { in.keep = __builtin_amdgcn_raw_buffer_load_b128(kr, (unsigned)(((((long long)dir * p.CS + cm.member) * B + bc) * 32 + (juc & 31)) * 4), 0, AUX_SC1); ++nload; }
        else if (p.dhn) { in.keep = __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(p.dhn + ((long long)dir * B + bc) * H + juc)); ++nload; }
        if (ps < T) {
            in.dy = z4;
            if (has_dy) { in.dy = *reinterpret_cast<const f32x4*>(p.dy + ((long long)t * B + bc) * ldy + dir * H + juc); ++nload; }
            // (layout of the saved gates: see the forward kernel's epilogue)
            const int gs = p.saved_mm ? 32 : H;
            const float* sv = p.saved_mm ? p.saved + ((((long long)dir * T + t) * p.CS + cm.member) * B + bc) * 128 + (juc & 31)
                                         : p.saved + (((long long)dir * T + t) * B + bc) * 4 * H + juc;
            in.rg = *reinterpret_cast<const f32x4*>(sv);
            in.zg = *reinterpret_cast<const f32x4*>(sv + gs);
            in.ng = *reinterpret_cast<const f32x4*>(sv + 2 * gs);
            in.q = *reinterpret_cast<const f32x4*>(sv + 3 * gs);
            in.hp = *reinterpret_cast<const f32x4*>(p.y_ext + ((long long)slot_prev * B + bc) * ldy + dir * H + juc);
            nload += 5;
        } else {
            in.dy = z4; in.rg = z4; in.zg = z4; in.ng = z4; in.q = z4; in.hp = z4;
        }
        return nload;
    };
    // The same for the main loop, WITHOUT a branch around any load (LD = 1: a step ps in [1, T): 7 loads -- dy from y_ext's
    // address when the layer has no dy, then dropped; LD = 2: ps == T, the running gradient only).  A load under a branch leaves
    // the compiler with paths of different pending counts: it then guards every later register write with a wait counted for the
    // shortest path, which on the long one waits for the stores issued in between (section 4.5.6 of DESIGN.md).
    auto epi_load_t = [&](auto LD, int ps, int r, EpiIn& in) -> int {
        constexpr int ld = decltype(LD)::value;
        const int s = T - 1 - ps;
        const int t = (dir == 0) ? s : T - 1 - s;
        const int slot_prev = (dir == 0) ? t : t + 2;
        const int b = m_base + 32 * r + gtr;
        const int bc = b < B ? b : B - 1;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        in.keep = __builtin_amdgcn_raw_buffer_load_b128(kr, (unsigned)(((((long long)dir * p.CS + cm.member) * B + bc) * 32 + (juc & 31)) * 4), 0, AUX_SC1);
        if constexpr (ld == 1) {
            const float* hpp = p.y_ext + ((long long)slot_prev * B + bc) * ldy + dir * H + juc;
            const f32x4 dyv = *reinterpret_cast<const f32x4*>(has_dy ? p.dy + ((long long)t * B + bc) * ldy + dir * H + juc : hpp);
            const int gs = p.saved_mm ? 32 : H;
            const float* sv = p.saved_mm ? p.saved + ((((long long)dir * T + t) * p.CS + cm.member) * B + bc) * 128 + (juc & 31)
                                         : p.saved + (((long long)dir * T + t) * B + bc) * 4 * H + juc;
            in.rg = *reinterpret_cast<const f32x4*>(sv);
            in.zg = *reinterpret_cast<const f32x4*>(sv + gs);
            in.ng = *reinterpret_cast<const f32x4*>(sv + 2 * gs);
            in.q = *reinterpret_cast<const f32x4*>(sv + 3 * gs);
            in.hp = *reinterpret_cast<const f32x4*>(hpp);
            in.dy = has_dy ? dyv : z4;
            return 7;
        } else {
            in.dy = z4; in.rg = z4; in.zg = z4; in.ng = z4; in.q = z4; in.hp = z4;
            return 1;
        }
    };
    // gate gradients of processing step ps for (round r, this lane's trial and units); acc = dgh_{ps-1} W_hh (own units).
    // Stores: the exchange rows first (group B), then the outputs (group C, always issued: dead lanes are dropped by the range
    // check); returns the number of group-C stores.
    // KIND (compile time; 0: decided at run time): 1 = exchange rows + outputs (a step that is contracted further), 3 = outputs only
    // (the last step when no dh0 is wanted), 2 = the dh0 row (ps == T)
    auto epilogue_t = [&](auto KIND, int ps, int r, const EpiIn& in, const f32x4& acc) -> int {
        constexpr int kind = decltype(KIND)::value;
        const int s = T - 1 - ps;
        const int t = (dir == 0) ? s : T - 1 - s;
        const int b = m_base + 32 * r + gtr;
        const bool live = b < B && ulive;
        f32x4 carry = __builtin_bit_cast(f32x4, in.keep);             // (whole vector: a bit_cast of ONE element reads element 0)
        if (kind != 0 || ps > 0) carry += acc;
        if (kind == 2 || (kind == 0 && ps == T)) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, carry), hr, live ? (unsigned)((((long long)dir * B + b) * H + ju) * 4) : CL_OOB, 0, 0);
            return 1;
        }
        f32x4 dar, daz, dan, danr, keep;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dh = in.dy[i] + carry[i];
            const float r_ = in.rg[i], z_ = in.zg[i], n_ = in.ng[i];
            const float dn = dh * (1.f - z_);
            const float dz = dh * (in.hp[i] - n_);
            const float da = dn * (1.f - n_ * n_);
            daz[i] = live ? dz * z_ * (1.f - z_) : 0.f;
            dar[i] = live ? da * in.q[i] * r_ * (1.f - r_) : 0.f;
            dan[i] = da;
            danr[i] = live ? da * r_ : 0.f;
            keep[i] = dh * z_;
        }
        const bool contracted = kind == 1 || (kind == 0 && ps + 1 < p.ps_total);      // someone will contract these gradients
        if (XOUT && contracted) {
            // the outputs dgi (r, z) and dghn ARE the exchange rows: issued first (the counted wait in front of the hand-off barrier
            // covers them), write-through unless the cluster sits on one XCD; dgi's n column and the running gradient follow below
            const unsigned go_ = live ? (unsigned)(((((long long)dir * T + t) * B + b) * 3 * H + ju) * 4) : CL_OOB;
            const unsigned no_ = live ? (unsigned)(((((long long)dir * T + t) * B + b) * H + ju) * 4) : CL_OOB;
            const u32x4 xr_ = __builtin_bit_cast(u32x4, split4_pack(dar)), xz_ = __builtin_bit_cast(u32x4, split4_pack(daz)),
                        xn_ = __builtin_bit_cast(u32x4, split4_pack(danr));
            if (fast) {
                __builtin_amdgcn_raw_buffer_store_b128(xr_, gr, go_, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(xz_, gr, live ? go_ + (unsigned)H * 4u : CL_OOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(xn_, nr, no_, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(xr_, gr, go_, 0, AUX_SC1);
                __builtin_amdgcn_raw_buffer_store_b128(xz_, gr, live ? go_ + (unsigned)H * 4u : CL_OOB, 0, AUX_SC1);
                __builtin_amdgcn_raw_buffer_store_b128(xn_, nr, no_, 0, AUX_SC1);
            }
            CL_FENCE();
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, split4_pack(dan)), gr, live ? go_ + (unsigned)H * 8u : CL_OOB, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, keep), kr, live ? (unsigned)(((((long long)dir * p.CS + cm.member) * B + b) * 32 + (ju & 31)) * 4) : CL_OOB, 0, 0);
            return 2;
        }
        if (!XOUT && contracted) {
            const unsigned chunk0 = (unsigned)((((ps & 1) * p.ndir + dir) * (p.Bp / 32) + (m_base / 32) + r) * 3);
            const unsigned rowoff = (unsigned)gtr * (unsigned)(KP * 4);
            const f32x4* gsrc[3] = {&dar, &daz, &danr};
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const unsigned base = (chunk0 + g) * (unsigned)Cf::CHUNK_BYTES + rowoff;
                if constexpr (BF) {
                    bf16x4 sh, sl;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { __bf16 a, c; bf_split((*gsrc[g])[i], a, c); sh[i] = a; sl[i] = c; }
                    if (XPS_CL_BWD_LINEPLANAR) {
                        // LINE-PLANAR row (alternative, compile time): every 128-byte line holds the hi halves of its 32 k (64 B), then
                        // their lo halves (64 B).  The two lanes of a unit octet swap one half (DPP) so that the even lane stores
                        // hi[0..7] and the odd lane lo[0..7] as ONE 16-byte vector each: three stores per lane and round instead of
                        // six, a member's 32 units of a gate = one FULL line per trial (the planes: two half lines), and a fragment
                        // register is still one contiguous 16-byte read (no regrouping in the contraction waves).  Bit-identical,
                        // all cluster tests green -- and SLOWER on one box, three interleaved runs each (tools/bench_gru.py with
                        // XPS_LIB_OVERRIDE): BPTT launch 812 / 812 / 829 us against 780 / 778 / 766 us with the planes.  The third
                        // exchange layout with 16-byte stores that loses 5-8 % in this kernel (XPS_CL_FWD_PLANES above).
                        const bool odd = (guq & 1) != 0;
                        const u32x2 mine_h = __builtin_bit_cast(u32x2, sh), mine_l = __builtin_bit_cast(u32x2, sl);
                        const u32x2 give = odd ? mine_h : mine_l;
                        u32x2 got;
                        got[0] = (unsigned)__shfl_xor((int)give[0], 1);
                        got[1] = (unsigned)__shfl_xor((int)give[1], 1);
                        const u32x4 v = odd ? (u32x4){got[0], got[1], mine_l[0], mine_l[1]} : (u32x4){mine_h[0], mine_h[1], got[0], got[1]};
                        const unsigned xo = base + ((unsigned)ju & ~31u) * 4u + (odd ? 64u : 0u) + (unsigned)(guq >> 1) * 16u;
                        if (fast) __builtin_amdgcn_raw_buffer_store_b128(v, xr, xo, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(v, xr, xo, 0, AUX_SC1);
                    } else if (fast) {
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sh), xr, base + (unsigned)ju * 2u, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sl), xr, base + KP * 2 + (unsigned)ju * 2u, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sh), xr, base + (unsigned)ju * 2u, 0, AUX_SC1);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sl), xr, base + KP * 2 + (unsigned)ju * 2u, 0, AUX_SC1);
                    }
                } else {
                    if (fast) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *gsrc[g]), xr, base + (unsigned)ju * 4u, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, *gsrc[g]), xr, base + (unsigned)ju * 4u, 0, AUX_SC1);
                }
            }
        }
        CL_FENCE();
        const unsigned go = live ? (unsigned)(((((long long)dir * T + t) * B + b) * 3 * H + ju) * 4) : CL_OOB;
        if (BF && p.split_out) {        // the only readers are GEMMs: hand them the hi / lo split instead of the fp32 values (same 16 bytes)
            dar = split4_pack(dar); daz = split4_pack(daz); dan = split4_pack(dan); danr = split4_pack(danr);
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dar), gr, go, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, daz), gr, live ? go + (unsigned)H * 4u : CL_OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dan), gr, live ? go + (unsigned)H * 8u : CL_OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, danr), nr, live ? (unsigned)(((((long long)dir * T + t) * B + b) * H + ju) * 4) : CL_OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, keep), kr, live ? (unsigned)(((((long long)dir * p.CS + cm.member) * B + b) * 32 + (ju & 31)) * 4) : CL_OOB, 0, 0);
        return 5;
    };
    auto epilogue = [&](int ps, int r, const EpiIn& in, const f32x4& acc) -> int {
        return epilogue_t(std::integral_constant<int, 0>{}, ps, r, in, acc);
    };

    int ps0 = p.ps_begin;
    if (ps0 == 0) {
        // first processing step: no contraction, the running gradient starts from dhn (or zero); gate waves only
        if (wave >= 4) {
            for (int r = 0; r < NR; ++r) {
                EpiIn in;
                epi_load(0, r, in);
                epilogue(0, r, in, (f32x4){0.f, 0.f, 0.f, 0.f});
            }
        }
        ps0 = 1;
        if (ps0 >= p.ps_end) return;
        // drain, then publish every round of step 0 at once
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int r = 0; p.handoff && wave == 4 && r < NR; r += 64)
            if (r + lane < NR) __hip_atomic_store(myflags + (r + lane) * 16 + cm.member, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    const int q_begin = ps0 * NQ, q_end = p.ps_end * NQ;
    if (p.handoff && wave == 4) {
        // the first two sub-iterations (round 0, segments 0 and 1) are loaded without a look-ahead poll
        unsigned f0 = 0xffffffffu;
        if (lane < p.CS) f0 = __hip_atomic_load(myflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cl_wait(myflags, (unsigned)ps0, p.CS, lane, f0, p.status, p.sticky);
    }
    __syncthreads();
    if (wave < 4) {
        const unsigned char* src = dma_src(ps0, 0, 0);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) dma_group(src, q_begin & 1, g4, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    if (wave < 4) {
        // ---------------- contraction waves ----------------
        const int ut = wave >> 1, kh = wave & 1;
        const int j0 = cm.member * Cf::U + ut * 16;
        const int kbase = kh * 256;
        const float* __restrict__ WT = p.w_hh_t[dir];
        ClWeights<BF> w;
        {
            const int jr = j0 + n;
            const bool rlive = jr < H;
            const float* wrow = WT + (long long)(rlive ? jr : 0) * 3 * H + kbase;
            const float* const seg[3] = {wrow, wrow + H, wrow + 2 * H};
            w.load(seg, rlive, kq, H - kbase, WT);
        }
        f32x4 acc[2];
        int ps = ps0, r = 0, g = 0;                      // (ps, r, g) of sub-iteration q
#ifdef XPS_CL_STAMP
        unsigned long long sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0, s_bar = 0, s_work = 0, s_drain = 0;
        CL_STAMP(sb2)
#endif
        for (int q = q_begin; q < q_end; ++q) {
            int ps_n = ps, r_n = r, g_n = g + 1;         // sub-iteration q + 1
            if (g_n == 3) { g_n = 0; if (++r_n == NR) { r_n = 0; ++ps_n; } }
            const bool has_next = q + 1 < q_end;
            const unsigned char* src = dma_src(ps_n, r_n, g_n);
            if (g == 0) {
                acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[1] = acc[0];
            }
            const unsigned char* tb = smem + (q & 1) * TILE;
            auto contract = [&](auto G) {
                constexpr int gg = decltype(G)::value;
                // one trial tile at a time (8 fragment registers live instead of 16: the kernel must fit 256 registers)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    if constexpr (BF) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            // (all four groups before the first chunk instead: 1120 -> 1197 us per launch)
                            // three of this wave's four groups; gate wave 4 + wave moves the fourth (stamps: issuing 16 pieces cost a
                            // contraction wave ~1300 cycles of a 3400-cycle sub-iteration while the gate waves idled at the barrier)
                            // (two groups on gate waves 5-7 as well: 991 vs 992 us, no further gain)
                            // (gg == 0: the gate waves are busy with the gate math and move one group; otherwise they idle and move two)
                            if (has_next && (c & 3) == 0 && tt * 2 + (c >> 2) < (gg == 0 ? 3 : 2)) dma_group(src, (q + 1) & 1, tt * 2 + (c >> 2), g_n);
                            // (planes: hi plane at the row, lo plane PS further; line-planar rows: k-half kh is its own piece, PS
                            //  apart, and a line holds hi[32] | lo[32] -- the same bank pattern per instruction)
                            bf16x8 bh, bl;
                            if constexpr (XOUT) {
                                // split4 groups, even | odd per line (cl_ximg_lane): groups 2 kq and 2 kq + 1 of chunk c, halves regrouped
                                const unsigned char* rp = tb + (tt * 16 + n) * TS + kh * PS + c * 128 + kq * 16;
                                const bf16x8 ga = *reinterpret_cast<const bf16x8*>(rp);
                                const bf16x8 gb = *reinterpret_cast<const bf16x8*>(rp + 64);
                                bh = __builtin_shufflevector(ga, gb, 0, 1, 2, 3, 8, 9, 10, 11);
                                bl = __builtin_shufflevector(ga, gb, 4, 5, 6, 7, 12, 13, 14, 15);
                            } else {
                                const unsigned char* rp = XPS_CL_BWD_LINEPLANAR ? tb + (tt * 16 + n) * TS + kh * PS + c * 128 + kq * 16
                                                                                : tb + (tt * 16 + n) * TS + (kbase + 32 * c + 8 * kq) * 2;
                                bh = *reinterpret_cast<const bf16x8*>(rp);
                                bl = *reinterpret_cast<const bf16x8*>(rp + (XPS_CL_BWD_LINEPLANAR ? 64 : PS));
                            }
                            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.wl[gg][c], bh, acc[tt], 0, 0, 0);
                            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.wh[gg][c], bl, acc[tt], 0, 0, 0);
                            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.wh[gg][c], bh, acc[tt], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 16; ++c) {
                            if (has_next && (c & 7) == 0) dma_group(src, (q + 1) & 1, tt * 2 + (c >> 3), g_n);
                            const f32x4 a4 = *reinterpret_cast<const f32x4*>(tb + (tt * 16 + n) * TS + (kbase + 16 * c + 4 * kq) * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.wf[gg][c][e], a4[e], acc[tt], 0, 0, 0);
                        }
                    }
                }
            };
            if (g == 0) contract(std::integral_constant<int, 0>{});
            else if (g == 1) contract(std::integral_constant<int, 1>{});
            else contract(std::integral_constant<int, 2>{});
            if (g == 2) {
                // (double buffered by round parity: the gate waves read it during the next sub-iteration, the next write
                // of the same buffer is six barriers away)
                unsigned char* xw = xacc + ((((q / 3) & 1) * 4 + wave) * 2) * 1024 + lane * 16;
                *reinterpret_cast<f32x4*>(xw) = acc[0];
                *reinterpret_cast<f32x4*>(xw + 1024) = acc[1];
            }
            ps = ps_n; r = r_n; g = g_n;
            CL_STAMP(sb3)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces have landed
            CL_STAMP(sb0)
            __syncthreads();
            CL_STAMP(sb1)
            CL_ACC(s_bar, sb0, sb1) CL_ACC(s_work, sb2, sb3) CL_ACC(s_drain, sb3, sb0)
#ifdef XPS_CL_STAMP
            sb2 = sb1;
#endif
        }
#ifdef XPS_CL_STAMP
        if (lane == 0) { const int wid = (blockIdx.x * 8 + wave) & 2047; g_clstamp[wid * 8 + 0] = s_work; g_clstamp[wid * 8 + 1] = s_drain; g_clstamp[wid * 8 + 2] = s_bar; }
#endif
        return;
    }

    // ---------------- gate waves ----------------
    __builtin_amdgcn_s_setprio(3);   // (see the forward kernel)
    EpiIn ein;
    int pend_ps = -1, pend_r = 0;    // round whose contraction is complete and whose gate math is due
    auto finish = [&](auto KIND, int qlast) -> int {   // qlast: the g == 2 sub-iteration of the pending round; returns the group-C stores
        // products of contraction wave (ut, kh), trial tile te: 1 KiB, lane (n, kq) of the MFMA layout at (16 kq + n) x 16 B
        const int ut_ = guq >> 2, te_ = gtr >> 4;
        const unsigned char* xa = xacc + (((qlast / 3) & 1) * 4) * 2 * 1024 + (((guq & 3) * 16 + (gtr & 15)) * 16);
        f32x4 a = *reinterpret_cast<const f32x4*>(xa + ((2 * ut_) * 2 + te_) * 1024);
        a += *reinterpret_cast<const f32x4*>(xa + ((2 * ut_ + 1) * 2 + te_) * 1024);
        return epilogue_t(KIND, pend_ps, pend_r, ein, a);
    };
#ifdef XPS_CL_STAMP
    unsigned long long sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0, sb4 = 0, s_bar = 0, s_work = 0, s_drain = 0, s_poll = 0;
    CL_STAMP(sb2)
#endif
    // One round = three sub-iterations (one barrier each), written out straight-line and instantiated per case -- FIN: the gate
    // math due at g == 0 (0: none, the launch's first round; 1 / 3 / 2: see epilogue_t),
    // LD: the inputs requested at g == 0 (see epi_load_t) -- so that no load and no store of the steady state sits under a branch.
    //   g == 0: [DMA] gate math of the round that finished at the last barrier (exchange rows, outputs), this round's inputs,
    //           counted wait (exchange rows + pieces complete)          g == 1: [DMA] flag, look-ahead poll          g == 2: [DMA]
    auto round_body = [&](auto FIN, auto LD, int ps, int r) {
        constexpr int fin = decltype(FIN)::value;
        const int q0 = ps * NQ + 3 * r;
        int r_n = r + 1, ps_n = ps;                           // the next round
        if (r_n == NR) { r_n = 0; ++ps_n; }
        const bool has_next = q0 + 3 < q_end;
        // ---------------- g == 0 ----------------
        if constexpr (BF) {
            // this wave's share of the next sub-iteration's operand: the fourth group of contraction wave (wave - 4)'s quarter,
            // first in the stream so that the counted wait covers it
            dma_group(dma_src(ps, r, 1), (q0 + 1) & 1, 3, 1);
            CL_FENCE();
        }
        int younger = 0;                                       // operations issued after the exchange stores (see cl_wait_vmcnt)
        if constexpr (fin != 0) younger += finish(FIN, q0 - 1);
        CL_FENCE();
        younger += epi_load_t(LD, ps, r, ein);                 // this round's inputs, used one round later
        CL_FENCE();
        CL_STAMP(sb3)
        cl_wait_vmcnt(younger);                                // exchange rows complete; outputs / inputs stay in flight
        CL_STAMP(sb4)
        CL_ACC(s_work, sb2, sb3) CL_ACC(s_drain, sb3, sb4)
        CL_STAMP(sb0)
        __syncthreads();
        CL_STAMP(sb1)
        CL_ACC(s_bar, sb0, sb1)
#ifdef XPS_CL_STAMP
        sb2 = sb1;
#endif
        // ---------------- g == 1 ----------------
        if constexpr (BF) {
            dma_group(dma_src(ps, r, 2), (q0 + 2) & 1, 2, 2);
            dma_group(dma_src(ps, r, 2), (q0 + 2) & 1, 3, 2);
            CL_FENCE();
        }
        // flag of the round whose gate math ran above: its exchange rows were complete before the last barrier
        if constexpr (fin != 0) {
            if (p.handoff && wave == 4 && lane == 0)
                __hip_atomic_store(myflags + pend_r * 16 + cm.member, (unsigned)(pend_ps + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // flags of the next round (its first operand image is requested during g == 2)
        const bool do_poll = p.handoff && wave == 4 && has_next;
        unsigned fl = 0xffffffffu;
        if (do_poll && lane < p.CS) fl = __hip_atomic_load(myflags + r_n * 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        CL_STAMP(sb3)
        if (do_poll) cl_wait(myflags + r_n * 16, (unsigned)ps_n, p.CS, lane, fl, p.status, p.sticky);
        if constexpr (BF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces have landed
        CL_STAMP(sb0)
        __syncthreads();
        CL_STAMP(sb1)
        CL_ACC(s_bar, sb0, sb1) CL_ACC(s_poll, sb3, sb0) CL_ACC(s_work, sb2, sb3)
#ifdef XPS_CL_STAMP
        sb2 = sb1;
#endif
        // ---------------- g == 2 ----------------
        if constexpr (BF) {
            if (has_next) { dma_group(dma_src(ps_n, r_n, 0), (q0 + 3) & 1, 2, 0); dma_group(dma_src(ps_n, r_n, 0), (q0 + 3) & 1, 3, 0); }
            CL_FENCE();
        }
        pend_ps = ps; pend_r = r;
        CL_STAMP(sb3)
        if constexpr (BF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CL_STAMP(sb0)
        __syncthreads();
        CL_STAMP(sb1)
        CL_ACC(s_bar, sb0, sb1) CL_ACC(s_poll, sb3, sb0) CL_ACC(s_work, sb2, sb3)
#ifdef XPS_CL_STAMP
        sb2 = sb1;
#endif
    };
    {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        bool first = true;
        for (int ps = ps0; ps < p.ps_end; ++ps) {
            for (int r = 0; r < NR; ++r) {
                const int pps = r == 0 ? ps - 1 : ps;              // step of the round whose gate math is due
                const int fk = first ? 0 : (pps == T ? 2 : (pps + 1 < p.ps_total ? 1 : 3));
                if (ps < T) {
                    if (fk == 1) round_body(I1{}, I1{}, ps, r);
                    else if (fk == 3) round_body(I3{}, I1{}, ps, r);
                    else round_body(I0{}, I1{}, ps, r);
                } else {
                    if (fk == 2) round_body(I2{}, I2{}, ps, r);
                    else if (fk == 1) round_body(I1{}, I2{}, ps, r);
                    else round_body(I0{}, I2{}, ps, r);
                }
                first = false;
            }
        }
#ifdef XPS_CL_STAMP
        if (lane == 0) { const int wid = (blockIdx.x * 8 + wave) & 2047; g_clstamp[wid * 8 + 0] = s_work; g_clstamp[wid * 8 + 1] = s_drain; g_clstamp[wid * 8 + 2] = s_bar; g_clstamp[wid * 8 + 3] = s_poll; }
#endif
    }
    if (pend_ps >= 0) finish(std::integral_constant<int, 0>{}, q_end - 1);                          // (outputs of the launch's last round; nobody waits for its flag)
}

// ------------------------------------------------------------------------------------------------------
// backward, TWO-DIMENSIONAL cluster (bf16x3 mode, 384 < H <= 512): opt-in alternative to the kernel above (see cl_plan2).
//
// The kernel above gives every member ALL 3H gate-gradient columns of its trials: 192 KiB of operand image per round of 32
// trials through the vector-memory path of every CU (1.5 MB per step), three barriers and 192 LDS-DMA pieces per round.
// Here the 16 members form a 4 x 4 grid (jg, kg): member (jg, kg) keeps W_hh^T[units of group jg (128)][gate-gradient columns
// of the units of group kg (3 x 128)] in registers (the same 196 KB) and needs only the K-SLICE kg of the gate gradients: 24 KiB
// per round of 16 trials, ONE image, one barrier.  What it computes is a PARTIAL sum over its slice for the 128 units of group
// jg; contraction wave w holds exactly the 32-unit quarter that member (jg, w) finishes, so it sends its accumulators straight
// to that member (2 KiB, global memory / L2) and the gate waves of (jg, w) add the four quarters in a FIXED order.  A third of
// the ingest of the 1-D kernel, at the price of a second hand-off per step.
//
//   iteration i = (ps - 1) * NR + r  (processing step ps >= 1, round r of 16 trials); slot s of the main loop (one barrier):
//     contraction waves : LDS-DMA of image s + 2 (ring of four), flag of the quarters of s - 1 (each wave for its own stores,
//                         behind a counted wait), MFMAs of iteration s with a pinned three-deep fragment prefetch, quarters out
//                         (own quarter: LDS ring)
//     gate waves        : a round of 16 trials x 32 units is 128 lanes of work: the gate waves form two PAIRS that take the
//                         iterations alternately, so every wave has two slots per iteration and each of its memory round
//                         trips a whole slot to complete in:  slot i + 1: saved gates / dy / h_prev / running gradient of
//                         iteration i requested (HBM);  slot i + 2: quarter flags, quarters requested (L2);  slot i + 3: gate
//                         math, exchange rows, outputs;  slot i + 4: the wave flags its exchange rows (they are the oldest
//                         operations still counted: the wait returns at once).  Wave 4 also looks the flags of image s + 3 up.
//   NR >= 16 rounds per cluster and step: the whole chain (contraction i -> quarters -> gate math -> gate gradients -> image
//   i + NR) spans 14 of 16 slots at most; lookups made a slot after publication hit unless the cluster is badly skewed.
//   Protocol as above (sc1 loads, write-through or same-XCD write-back stores, counted vmcnt, flags stored by a lane of the
//   storing wave / polled by the loading wave or in front of a barrier); bounded spins.
//   One step per launch (XPS_GRU_CLUSTER=steps, or a grid the device cannot hold at once): the same kernel twice per step --
//   contraction only, then gate math only -- with every quarter passed through global memory: same sums, same bits.
//
// Operand image: [trial 16][plane 2][384] bf16, trials 1536 B apart WITHOUT padding; 16-byte chunk q of trial n sits at chunk
// q ^ n of its 256-byte window -- applied by the WRITERS of the gate gradients (the LDS-DMA is a linear copy: a piece whose
// lanes fetch permuted chunks is split into per-lane requests and blocks its issuer ~650 instead of ~100 cycles), undone by
// the fragment reads -- which makes the b128 reads of every 16-lane group hit 16 distinct bank quads.
// ------------------------------------------------------------------------------------------------------
struct ClBwd2 {
    const float* dy;
    const float* dhn;
    const float* y_ext;
    const float* saved;
    const float* w_hh_t[2];     // (H x 3H)
    float* dgi;
    float* dghn;
    float* dh0;
    float* keep;                // [ndir][B][H]  z * dh of the step processed before
    void* xbuf;                 // [2 parity][ndir][Bp/16][4 slices][16 trials][2 planes][384] bf16 (swizzled chunks)
    float* pbuf;                // [2 parity][cluster][NR][16 dst][4 src][512] partial quarters
    unsigned* flags;            // [cluster][NR][160]: [member 16][gate wave 4] "gate gradients of step v - 1 published", then
                                // [dst 16][src 4] "quarter of step v published", pad
    unsigned* xcc;
    unsigned* status;
    unsigned* sticky;
    unsigned xbuf_bytes, pbuf_bytes;
    int T, B, H, ndir, Bp, Mc, NR, nblk;
    int ps_total, handoff, do_ps0;
    int c_begin, c_end, g_begin, g_end;     // iterations whose contraction / gate math this launch runs
    int saved_mm;               // saved gates member-major (see the forward kernel's epilogue)
    int split_out;
};

#if defined(XPS_CL2_NT_LOADS)
#define CL2_STREAM_LOAD(ptr) __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ptr))
#else
#define CL2_STREAM_LOAD(ptr) (*reinterpret_cast<const f32x4*>(ptr))
#endif
#if defined(XPS_CL2_NT_STORES)
constexpr int CL2_AUX_STREAM = 2;           // raw buffer builtins: bit 1 = nt
#elif defined(XPS_CL2_SC1_STORES)
constexpr int CL2_AUX_STREAM = 16;          // write-through, line dropped from L2
#else
constexpr int CL2_AUX_STREAM = 0;
#endif
constexpr int C2_RT = 16;                   // trials per round
constexpr int C2_IMG = C2_RT * 1536;        // one operand image
constexpr int C2_NIMG = 4;                  // image ring
constexpr int C2_LDS = C2_NIMG * C2_IMG + 4 * 2048 + 64;

// two 1-KiB LDS-DMA pieces (see cl_dma4)
__device__ inline void cl_dma2(const unsigned char* gsrc, unsigned l0) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024 sc1\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(l0));
}

__global__ __launch_bounds__(512, 2) void gru_cluster2_bwd_kernel(ClBwd2 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ownq = smem + C2_NIMG * C2_IMG;     // ring of four own-quarter tiles (2 KiB each)
    const unsigned lds0 = cl_lds_base(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kq = lane >> 4;
    const ClMap cm = cl_map(16);
    const int dir = cm.cluster / p.nblk, blk = cm.cluster % p.nblk;
    const int jg = cm.member >> 2, kg = cm.member & 3;
    const int T = p.T, B = p.B, H = p.H, NR = p.NR;
    const int ldy = p.ndir * H;
    const int m_base = blk * p.Mc;
    const int nclusters = p.ndir * p.nblk;
    unsigned* myflags = p.flags + (long long)cm.cluster * NR * 160;
    const bool fast = !p.handoff || cl_same_xcd(p.xcc + cm.cluster * 16, cm.member, 16, lane, wave == 4, p.status, p.sticky, reinterpret_cast<unsigned*>(smem));
    const bool has_c = p.c_end > p.c_begin, has_g = p.g_end > p.g_begin;
    auto valid_c = [&](int i) { return i >= p.c_begin && i < p.c_end; };
    auto valid_g = [&](int i) { return i >= p.g_begin && i < p.g_end; };
    int s_lo = 0x7fffffff, s_hi = -0x7fffffff;
    if (has_c) { s_lo = p.c_begin; s_hi = p.c_end - 1; }
    if (has_g) { s_lo = min(s_lo, p.g_begin + 1); s_hi = max(s_hi, p.g_end + 3); }

    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(p.xbuf, 0, p.xbuf_bytes, RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(p.pbuf, 0, p.pbuf_bytes, RSRC_FLAGS);
    // byte offset of the quarter tile (parity, round, destination member, source column) in pbuf
    auto p_off = [&](int ps_, int r_, int dst, int src) -> unsigned {
        return (unsigned)((((((ps_ & 1) * nclusters + cm.cluster) * NR + r_) * 16 + dst) * 4 + src)) * 2048u;
    };
    // chunk (one slice of one round) of the gate-gradient exchange buffer
    auto x_chunk = [&](int ps_src, int r_, int slice) -> unsigned {
        return (unsigned)(((((ps_src & 1) * p.ndir + dir) * (p.Bp / C2_RT) + (m_base / C2_RT) + r_) * 4 + slice)) * (unsigned)C2_IMG;
    };
    // flags of round r: words [4 m, 4 m + 4) gate gradients of member m (one word per gate wave), words [64 + 4 dst, + 4) quarters
    auto flag_image = [&](int i) -> const unsigned* { return myflags + (i % NR) * 160 + kg * 16; };
    auto flag_quarters = [&](int i) -> const unsigned* { return myflags + (i % NR) * 160 + 64 + cm.member * 4; };
    // (skip_own: the four quarter flags of this member, its own column excluded; else: the 16 gate-gradient flags -- four gate
    //  waves of each of the four members that write the slice)
    auto peek = [&](const unsigned* f, bool skip_own) -> unsigned {
        unsigned v = 0xffffffffu;
        if (skip_own ? (lane < 4 && lane != kg) : lane < 16) v = __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    };
    auto wait_flags = [&](const unsigned* f, bool skip_own, unsigned need, unsigned first, int stat) {
        if (__all(first >= need)) return;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool ok = false;
        for (;;) {
            const unsigned v = peek(f, skip_own);
            if (__all(v >= need)) { ok = true; break; }
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) break;
        }
        if (lane == 0) {
            const unsigned dt = (unsigned)(__builtin_amdgcn_s_memrealtime() - t0);
            if (!ok) { __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (p.sticky) __hip_atomic_store(p.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            __hip_atomic_fetch_add(p.status + stat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(p.status + stat + 1, dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    if (wave < 4) {
        // ---------------- contraction waves ----------------
        if (p.do_ps0) { __syncthreads(); }
        if (s_lo > s_hi) return;
        // LDS-DMA of one image: a linear copy of the 24-KiB chunk; wave w moves bytes [6 w, 6 w + 6) KiB
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.xbuf);
        auto dma_image = [&](int i) {
            const int ps = i / NR + 1, r = i % NR;
            const unsigned char* src = xb + x_chunk(ps - 1, r, kg) + wave * 6144 + lane * 16;
            const unsigned dst = lds0 + (unsigned)((i & (C2_NIMG - 1)) * C2_IMG) + (unsigned)(wave * 6144);
            cl_dma4(src, dst, dst, dst, dst);
            cl_dma2(src + 4096, dst + 4096);
        };
        if (has_c && p.handoff) __syncthreads();           // (wave 4's polls for the first three images)
        if (has_c) {
            dma_image(p.c_begin);
            if (valid_c(p.c_begin + 1)) dma_image(p.c_begin + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                   // the first images have landed
        const float* __restrict__ WT = p.w_hh_t[dir];
        bf16x8 wh[2][12], wl[2][12];
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) {
            const int jr = jg * 128 + wave * 32 + ut * 16 + n;
            const bool rlive = has_c && jr < H;                // (a gate-only launch contracts nothing: no weight traffic)
            const float* wrow = WT + (long long)(rlive ? jr : 0) * 3 * H;
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                const int ku = kg * 128 + (c & 3) * 32 + 8 * kq;
                const float* src = wrow + (c >> 2) * H + ku;
                const bool ok0 = rlive && ku + 3 < H, ok1 = rlive && ku + 7 < H;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(ok0 ? src : WT);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(ok1 ? src + 4 : WT);
                if (!ok0) v0 = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (!ok1) v1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    __bf16 a_, b_;
                    bf_split(v0[e], a_, b_); wh[ut][c][e] = a_; wl[ut][c][e] = b_;
                    bf_split(v1[e], a_, b_); wh[ut][c][4 + e] = a_; wl[ut][c][4 + e] = b_;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the weight loads: the counted waits below assume an empty queue)
        const bool own_lds = p.handoff && wave == kg;      // this wave's quarter is the member's own: through LDS
        // fragment addresses: chunk (4 c + kq) ^ n = 16 (c >> 2) + (4 ((c & 3) ^ (n >> 2)) | (kq ^ (n & 3))): four lane offsets (by
        // c & 3), the rest is an immediate
        unsigned frag_lane[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) frag_lane[m] = (unsigned)(n * 1536 + ((((m ^ (n >> 2)) << 2) | (kq ^ (n & 3))) << 4));
#ifdef XPS_CL_STAMP
        unsigned long long sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0, sbA = 0, sbB = 0, s_bar = 0, s_work = 0, s_drain = 0, s_dma = 0, s_mma = 0;
        CL_STAMP(sb2)
#endif
        for (int s = s_lo; s <= s_hi; ++s) {
            // (1) the image of iteration s + 2 (its flags were looked up by wave 4 before the last barrier)
            const bool dma_now = valid_c(s + 2);
            if (dma_now) dma_image(s + 2);
            CL_FENCE();
            // (2) the quarters stored at the end of the last slot are in memory by now (they had the barrier and the DMA issue):
            //     this wave flags them for their reader
            if (p.handoff && !own_lds && valid_c(s - 1)) {
                if (dma_now) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    const int i = s - 1;
                    __hip_atomic_store(myflags + (i % NR) * 160 + 64 + (jg * 4 + wave) * 4 + kg, (unsigned)(i / NR + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            CL_FENCE();
            CL_STAMP(sbA)
            if (valid_c(s)) {
                const int ps = s / NR + 1, r = s % NR;
                f32x4 acc[2];
                acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[1] = acc[0];
                // 12 steps (32-wide k chunks); the fragments of step i + 2 are requested before the MFMAs of step i (a ring of
                // three, pinned with scheduling barriers: left alone the scheduler issues every read right in front of its six
                // MFMAs and exposes the LDS latency every time)
                unsigned fa[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) fa[m] = frag_lane[m] + (unsigned)((s & (C2_NIMG - 1)) * C2_IMG);
                bf16x8 fh[3], fl[3];
                auto frag = [&](int i, bf16x8& h8, bf16x8& l8) {
                    const unsigned char* rp = smem + fa[i & 3] + (i >> 2) * 256;
                    h8 = *reinterpret_cast<const bf16x8*>(rp);
                    l8 = *reinterpret_cast<const bf16x8*>(rp + 768);
                };
                frag(0, fh[0], fl[0]);
                frag(1, fh[1], fl[1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    if (i + 2 < 12) frag(i + 2, fh[(i + 2) % 3], fl[(i + 2) % 3]);
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8 bh = fh[i % 3], bl = fl[i % 3];
#pragma unroll
                    for (int ut = 0; ut < 2; ++ut) {
                        acc[ut] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ut][i], bh, acc[ut], 0, 0, 0);
                        acc[ut] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ut][i], bl, acc[ut], 0, 0, 0);
                        acc[ut] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ut][i], bh, acc[ut], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                CL_STAMP(sbB)
                CL_ACC(s_mma, sbA, sbB)
                if (own_lds) {
#pragma unroll
                    for (int ut = 0; ut < 2; ++ut)
                        *reinterpret_cast<f32x4*>(ownq + (s & 3) * 2048 + (ut * 64 + lane) * 16) = acc[ut];
                } else {
                    const unsigned base = p_off(ps, r, jg * 4 + wave, kg) + (unsigned)lane * 16u;
#pragma unroll
                    for (int ut = 0; ut < 2; ++ut) {
                        if (fast) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[ut]), pr, base + (unsigned)ut * 1024u, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[ut]), pr, base + (unsigned)ut * 1024u, 0, AUX_SC1);
                    }
                }
            }
            CL_FENCE();
            CL_STAMP(sb3)
            // (3) the image of iteration s + 1 (requested a slot ago) has landed: everything older than this slot's operations
            //     (6 pieces, 1 flag, 2 stores -- those that were issued) is complete
            cl_wait_vmcnt((dma_now ? 6 : 0) + ((p.handoff && !own_lds && valid_c(s - 1)) ? 1 : 0) + ((valid_c(s) && !own_lds) ? 2 : 0));
            CL_STAMP(sb0)
            __syncthreads();
            CL_STAMP(sb1)
            CL_ACC(s_bar, sb0, sb1) CL_ACC(s_work, sb2, sb3) CL_ACC(s_drain, sb3, sb0) CL_ACC(s_dma, sb2, sbA)
#ifdef XPS_CL_STAMP
            sb2 = sb1;
#endif
        }
#ifdef XPS_CL_STAMP
        if (lane == 0) { const int wid = (blockIdx.x * 8 + wave) & 2047; g_clstamp[wid * 8 + 0] = s_work; g_clstamp[wid * 8 + 1] = s_drain; g_clstamp[wid * 8 + 2] = s_bar; g_clstamp[wid * 8 + 4] = s_dma; g_clstamp[wid * 8 + 5] = s_mma; }
#endif
        return;
    }

    // ---------------- gate waves ----------------
    __builtin_amdgcn_s_setprio(3);
    const int hw = wave - 4;
    const int pair = hw >> 1, hut = hw & 1;               // pair: takes the iterations i with i % 2 == pair; unit tile
    // lane (t8, uq): trial 8 hut + t8 of the round's 16, units 4 uq .. 4 uq + 3 of the member's 32: eight full 128-byte lines per
    // wave-instruction on every stream (see the 1-D kernel)
    const int gtr = 8 * hut + (lane >> 3);
    const int guq = lane & 7;
    const int uo = kg * 32 + 4 * guq;
    const unsigned qtile = (unsigned)((((guq >> 2) * 64) + (guq & 3) * 16 + gtr) * 16);    // this lane's products inside a quarter tile
    const int ju = jg * 128 + uo;
    const bool ulive = ju < H;
    const int juc = ulive ? ju : 0;
    __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(p.keep, 0, (unsigned)((long long)p.ndir * 16 * B * 32 * 4), RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(p.dgi, 0, (unsigned)((long long)p.ndir * T * B * 3 * H * 4), RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t nr = __builtin_amdgcn_make_buffer_rsrc(p.dghn, 0, (unsigned)((long long)p.ndir * T * B * H * 4), RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t hr = __builtin_amdgcn_make_buffer_rsrc(p.dh0, 0, p.dh0 ? (unsigned)((long long)p.ndir * B * H * 4) : 0u, RSRC_FLAGS);
    const bool has_dy = p.dy != nullptr;

    struct EpiIn { f32x4 dy, rg, zg, ng, q, hp; u32x4 keep; };
    auto epi_load = [&](int ps, int r, EpiIn& in) -> int {
        const int s_ = T - 1 - ps;
        const int t = (dir == 0) ? s_ : T - 1 - s_;
        const int slot_prev = (dir == 0) ? t : t + 2;
        const int b = m_base + C2_RT * r + gtr;
        const int bc = b < B ? b : B - 1;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        int nload = 0;
        in.keep = (u32x4){0u, 0u, 0u, 0u};
        if (ps > 0) { in.keep = __builtin_amdgcn_raw_buffer_load_b128(kr, (unsigned)(((((long long)dir * 16 + cm.member) * B + bc) * 32 + (juc & 31)) * 4), 0, AUX_SC1); ++nload; }
        else if (p.dhn) { in.keep = __builtin_bit_cast(u32x4, *reinterpret_cast<const f32x4*>(p.dhn + ((long long)dir * B + bc) * H + juc)); ++nload; }
        if (ps < T) {
            in.dy = z4;
            if (has_dy) { in.dy = CL2_STREAM_LOAD(p.dy + ((long long)t * B + bc) * ldy + dir * H + juc); ++nload; }
            const int gs = p.saved_mm ? 32 : H;
            const float* sv = p.saved_mm ? p.saved + ((((long long)dir * T + t) * ((H + 31) / 32) + cm.member) * B + bc) * 128 + (juc & 31)
                                         : p.saved + (((long long)dir * T + t) * B + bc) * 4 * H + juc;
            in.rg = CL2_STREAM_LOAD(sv);
            in.zg = CL2_STREAM_LOAD(sv + gs);
            in.ng = CL2_STREAM_LOAD(sv + 2 * gs);
            in.q = CL2_STREAM_LOAD(sv + 3 * gs);
            in.hp = CL2_STREAM_LOAD(p.y_ext + ((long long)slot_prev * B + bc) * ldy + dir * H + juc);
            nload += 5;
        } else {
            in.dy = z4; in.rg = z4; in.zg = z4; in.ng = z4; in.q = z4; in.hp = z4;
        }
        return nload;
    };
    // the same for an interior iteration (1 <= ps < T) without a branch around any load: 7 loads (see the 1-D kernel's epi_load_t)
    auto epi_load_fast = [&](int ps, int r, EpiIn& in) {
        const int s_ = T - 1 - ps;
        const int t = (dir == 0) ? s_ : T - 1 - s_;
        const int slot_prev = (dir == 0) ? t : t + 2;
        const int b = m_base + C2_RT * r + gtr;
        const int bc = b < B ? b : B - 1;
        in.keep = __builtin_amdgcn_raw_buffer_load_b128(kr, (unsigned)(((((long long)dir * 16 + cm.member) * B + bc) * 32 + (juc & 31)) * 4), 0, AUX_SC1);
        const float* hpp = p.y_ext + ((long long)slot_prev * B + bc) * ldy + dir * H + juc;
        const f32x4 dyv = CL2_STREAM_LOAD(has_dy ? p.dy + ((long long)t * B + bc) * ldy + dir * H + juc : hpp);
        const int gs = p.saved_mm ? 32 : H;
        const float* sv = p.saved_mm ? p.saved + ((((long long)dir * T + t) * ((H + 31) / 32) + cm.member) * B + bc) * 128 + (juc & 31)
                                     : p.saved + (((long long)dir * T + t) * B + bc) * 4 * H + juc;
        in.rg = CL2_STREAM_LOAD(sv);
        in.zg = CL2_STREAM_LOAD(sv + gs);
        in.ng = CL2_STREAM_LOAD(sv + 2 * gs);
        in.q = CL2_STREAM_LOAD(sv + 3 * gs);
        in.hp = CL2_STREAM_LOAD(hpp);
        in.dy = has_dy ? dyv : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    // gate gradients of processing step ps, round r for this lane's trial and four units; acc = dgh_{ps-1} W_hh (own units).
    // Stores: the exchange rows first, then the outputs (always issued: dead lanes are dropped by the range check).
    // KIND 1 (compile time): an interior step -- exchange rows + outputs, no branch; 0: decided at run time.
    struct EpiOut { f32x4 dar, daz, dan, danr, keep; unsigned go, no, ko; };
    auto output_rows = [&](const EpiOut& o) {              // five stores (dead lanes: out of range, dropped)
        const bool live = o.go != CL_OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o.dar), gr, o.go, 0, CL2_AUX_STREAM);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o.daz), gr, live ? o.go + (unsigned)H * 4u : CL_OOB, 0, CL2_AUX_STREAM);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o.dan), gr, live ? o.go + (unsigned)H * 8u : CL_OOB, 0, CL2_AUX_STREAM);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o.danr), nr, o.no, 0, CL2_AUX_STREAM);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o.keep), kr, o.ko, 0, 0);
    };
    // (defer != nullptr: the output rows are not stored but handed back, for output_rows() a slot later)
    auto epilogue_t = [&](auto KIND, int ps, int r, const EpiIn& in, const f32x4& acc, EpiOut* defer = nullptr) {
        constexpr int kind = decltype(KIND)::value;
        const int s_ = T - 1 - ps;
        const int t = (dir == 0) ? s_ : T - 1 - s_;
        const int b = m_base + C2_RT * r + gtr;
        const bool live = b < B && ulive;
        f32x4 carry = __builtin_bit_cast(f32x4, in.keep);
        if (kind != 0 || ps > 0) carry += acc;
        if (kind == 0 && ps == T) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, carry), hr, live ? (unsigned)((((long long)dir * B + b) * H + ju) * 4) : CL_OOB, 0, 0);
            return;
        }
        f32x4 dar, daz, dan, danr, keep;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dh = in.dy[i] + carry[i];
            const float r_ = in.rg[i], z_ = in.zg[i], n_ = in.ng[i];
            const float dn = dh * (1.f - z_);
            const float dz = dh * (in.hp[i] - n_);
            const float da = dn * (1.f - n_ * n_);
            daz[i] = live ? dz * z_ * (1.f - z_) : 0.f;
            dar[i] = live ? da * in.q[i] * r_ * (1.f - r_) : 0.f;
            dan[i] = da;
            danr[i] = live ? da * r_ : 0.f;
            keep[i] = dh * z_;
        }
        if (kind == 1 || ps + 1 < p.ps_total) {         // someone will contract these gradients: slice jg of round r
            // row of trial n, plane-row byte L = 2 (128 g + uo) -> 16-byte chunk (L >> 4) ^ n of its 256-byte window (the operand
            // image's bank swizzle, see the kernel header), byte L & 15 inside the chunk
            const unsigned base = x_chunk(ps, r, jg) + (unsigned)gtr * 1536u;
            const unsigned Lq = (unsigned)uo >> 3, Lw = ((unsigned)uo & 4u) << 1;
            const f32x4* gsrc[3] = {&dar, &daz, &danr};
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                bf16x4 sh, sl;
#pragma unroll
                for (int i = 0; i < 4; ++i) { __bf16 a_, c_; bf_split((*gsrc[g])[i], a_, c_); sh[i] = a_; sl[i] = c_; }
                const unsigned o = base + ((((unsigned)(16 * g) + Lq) ^ (unsigned)gtr) << 4) + Lw;
                if (fast) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sh), xr, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sl), xr, o + 768u, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sh), xr, o, 0, AUX_SC1);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, sl), xr, o + 768u, 0, AUX_SC1);
                }
            }
        }
        CL_FENCE();
        if (p.split_out) { dar = split4_pack(dar); daz = split4_pack(daz); dan = split4_pack(dan); danr = split4_pack(danr); }
        EpiOut o;
        o.dar = dar; o.daz = daz; o.dan = dan; o.danr = danr; o.keep = keep;
        o.go = live ? (unsigned)(((((long long)dir * T + t) * B + b) * 3 * H + ju) * 4) : CL_OOB;
        o.no = live ? (unsigned)(((((long long)dir * T + t) * B + b) * H + ju) * 4) : CL_OOB;
        o.ko = live ? (unsigned)(((((long long)dir * 16 + cm.member) * B + b) * 32 + (ju & 31)) * 4) : CL_OOB;
        if (defer) *defer = o;
        else output_rows(o);
    };
    auto epilogue = [&](int ps, int r, const EpiIn& in, const f32x4& acc) { epilogue_t(std::integral_constant<int, 0>{}, ps, r, in, acc); };

    if (p.do_ps0) {
        // first processing step: no contraction, the running gradient starts from dhn (or zero); pair P takes the rounds r % 2 == P
        for (int r = pair; r < NR; r += 2) {
            EpiIn in;
            epi_load(0, r, in);
            epilogue(0, r, in, (f32x4){0.f, 0.f, 0.f, 0.f});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int r = pair + 2 * lane; p.handoff && r < NR; r += 128)      // (every gate wave for its own stores; two words per member and round)
            for (int k = 0; k < 2; ++k)
                __hip_atomic_store(myflags + r * 160 + cm.member * 4 + 2 * k + hut, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    if (s_lo > s_hi) return;

    // the first three images: their flags (persistent form), then the contraction waves' LDS-DMA of two; barriers mirror theirs
    if (has_c && p.handoff) {
        if (wave == 4)
            for (int i = p.c_begin; i < p.c_begin + 3 && valid_c(i); ++i)
                wait_flags(flag_image(i), false, (unsigned)(i / NR + 1), 0u, 1);
        __syncthreads();
    }
    __syncthreads();

    // A round of 16 trials is the work of ONE pair; the other pair takes the next.  Wave of pair P, slot s:
    //   s % 2 != P ("math slot", i = s - 3 has i % 2 == P): gate math of iteration s - 3, then the HBM requests of iteration s - 1
    //   s % 2 == P ("request slot"): flag of its exchange rows of iteration s - 4, quarter flags and quarters of iteration s - 2
    // (iteration i: requests in slot i + 1 and i + 2, math in slot i + 3, flag in slot i + 4: two slots per memory round trip)
    EpiIn ein;
    f32x4 pq[4], pown;                                     // quarters from the other members (pq[kg]: nothing) / the own one (LDS)
#pragma unroll
    for (int c = 0; c < 4; ++c) pq[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    pown = pq[0];
    int after_rows = 0;                                    // operations issued behind the exchange rows of the last math slot
#ifdef XPS_CL_STAMP
    unsigned long long sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0, sb4 = 0, s_bar = 0, s_work = 0, s_drain = 0, s_poll = 0;
    CL_STAMP(sb2)
#endif
    const bool own_q = p.handoff != 0;                     // the member's own quarter comes through LDS
    // sum of the four quarters in the fixed order 0, 1, 2, 3 (the own one from its LDS copy)
    auto quarter_sum = [&]() -> f32x4 {
        const f32x4 q0 = (own_q && kg == 0) ? pown : pq[0], q1 = (own_q && kg == 1) ? pown : pq[1];
        const f32x4 q2 = (own_q && kg == 2) ? pown : pq[2], q3 = (own_q && kg == 3) ? pown : pq[3];
        return ((q0 + q1) + q2) + q3;
    };
    // requests for the quarters of iteration i: four loads without a branch (the own column's is sent out of range and dropped:
    // a load under a run-time branch got a vmcnt(0) apiece from the compiler), the own quarter from LDS
    auto quarter_requests = [&](int i) {
        const int ps = i / NR + 1, r = i % NR;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            pq[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pr, (own_q && c == kg) ? CL_OOB : p_off(ps, r, cm.member, c) + qtile, 0, AUX_SC1));
        pown = *reinterpret_cast<const f32x4*>(ownq + (i & 3) * 2048 + qtile);
    };
    // ---- one slot, general form (edges of the launch, the last step, the one-step-per-launch modes) ----
    // A round of 16 trials is the work of ONE pair; the other pair takes the next.  Wave of pair P, slot s:
    //   s % 2 != P ("math slot", i = s - 3 has i % 2 == P): gate math of iteration s - 3, then the HBM requests of iteration s - 1
    //   s % 2 == P ("request slot"): flag of its exchange rows of iteration s - 4, quarter flags and quarters of iteration s - 2
    // (iteration i: requests in slot i + 1 and i + 2, math in slot i + 3, flag in slot i + 4: two slots per memory round trip)
    auto slot_general = [&](int s) {
        unsigned la_img = 0xffffffffu;
        if (p.handoff && wave == 4 && valid_c(s + 3)) la_img = peek(flag_image(s + 3), false);
        CL_FENCE();
        if (((s - pair) & 1) != 0) {
            // ---- math slot ----
            int younger = 0;
            if (valid_g(s - 3)) {
                const int i = s - 3, ps = i / NR + 1;
                epilogue(ps, i % NR, ein, quarter_sum());
                younger += ps == T ? 1 : 5;
            }
            CL_FENCE();
            if (valid_g(s - 1)) younger += epi_load((s - 1) / NR + 1, (s - 1) % NR, ein);
            after_rows = younger;
            CL_STAMP(sb3)
        } else {
            // ---- request slot ----
            if (p.handoff && valid_g(s - 4)) {
                const int i = s - 4, ps = i / NR + 1;
                if (ps + 1 < p.ps_total) {
                    cl_wait_vmcnt(after_rows);               // the exchange rows of slot s - 1: the oldest operations still counted
                    if (lane == 0) {
                        unsigned* f = myflags + (i % NR) * 160 + cm.member * 4 + hut;
                        __hip_atomic_store(f, (unsigned)(ps + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(f + 2, (unsigned)(ps + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            CL_FENCE();
            if (valid_g(s - 2)) {
                const int i = s - 2;
                if (p.handoff) wait_flags(flag_quarters(i), true, (unsigned)(i / NR + 1), peek(flag_quarters(i), true), 6);
                quarter_requests(i);
            }
            CL_STAMP(sb3)
        }
        CL_FENCE();
        CL_STAMP(sb4)
        // the image of iteration s + 3 is DMA'd in the next slot: its flags must have matched before the barrier
        if (p.handoff && wave == 4 && valid_c(s + 3)) wait_flags(flag_image(s + 3), false, (unsigned)((s + 3) / NR + 1), la_img, 1);
        CL_STAMP(sb0)
        __syncthreads();
        CL_STAMP(sb1)
        CL_ACC(s_bar, sb0, sb1) CL_ACC(s_work, sb2, sb3) CL_ACC(s_drain, sb3, sb4) CL_ACC(s_poll, sb4, sb0)
#ifdef XPS_CL_STAMP
        sb2 = sb1;
#endif
    };
    // ---- a math slot s and the request slot s + 1 of this pair, interior of the persistent launch: straight-line ----
    // Every iteration touched (s - 3 .. s + 4) exists and lies before the last processing step, the cluster sits on one XCD.  No
    // load sits under a branch (the compiler guards register writes behind a branch join with waits counted for the shorter path:
    // behind eleven fresh stores they wait for the stores; the general form above loses ~1100 cycles per slot to them), every
    // wave looks the image flags up (only wave 4 acts on them), and the quarter flags of the request slot are looked up a slot
    // early, behind the stores of the math slot.
    auto slots_fast = [&](int s) {
        // ---- math slot s: gate math of iteration s - 3, requests of iteration s - 1 ----
        unsigned la_img = peek(flag_image(s + 3), false);
        CL_FENCE();
        {
            const int i = s - 3;
            epilogue_t(std::integral_constant<int, 1>{}, i / NR + 1, i % NR, ein, quarter_sum());
        }
        CL_FENCE();
        const unsigned lq = peek(flag_quarters(s - 1), true);     // consumed in the request slot
        CL_FENCE();
        epi_load_fast((s - 1) / NR + 1, (s - 1) % NR, ein);
        CL_FENCE();
        CL_STAMP(sb3)
        CL_STAMP(sb4)
        if (wave == 4) wait_flags(flag_image(s + 3), false, (unsigned)((s + 3) / NR + 1), la_img, 1);
        CL_STAMP(sb0)
        __syncthreads();
        CL_STAMP(sb1)
        CL_ACC(s_bar, sb0, sb1) CL_ACC(s_work, sb2, sb3) CL_ACC(s_drain, sb2, sb3)      /* (stamped build: s_drain = math slots, s_poll = request slots) */
#ifdef XPS_CL_STAMP
        sb2 = sb1;
#endif
        // ---- request slot s + 1: flag of the exchange rows of iteration s - 3, quarters of iteration s - 1 ----
        la_img = peek(flag_image(s + 4), false);
        CL_FENCE();
        cl_wait_vmcnt(5 + 1 + 7 + 1);                           // behind the exchange rows: 5 outputs, the quarter-flag look-up, 7 inputs, the image look-up
        if (lane == 0) {
            const int i = s - 3;
            unsigned* f = myflags + (i % NR) * 160 + cm.member * 4 + hut;
            __hip_atomic_store(f, (unsigned)(i / NR + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(f + 2, (unsigned)(i / NR + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        CL_FENCE();
        // (the output rows of the math slot stored here instead, to halve that slot's vector-memory work: 789 -> 1937 us -- the
        //  quarter requests then queue behind five HBM stores and every next math slot waits for them)
        wait_flags(flag_quarters(s - 1), true, (unsigned)((s - 1) / NR + 1), lq, 6);
        quarter_requests(s - 1);
        CL_FENCE();
        CL_STAMP(sb3)
        CL_STAMP(sb4)
        if (wave == 4) wait_flags(flag_image(s + 4), false, (unsigned)((s + 4) / NR + 1), la_img, 1);
        CL_STAMP(sb0)
        __syncthreads();
        CL_STAMP(sb1)
        CL_ACC(s_bar, sb0, sb1) CL_ACC(s_work, sb2, sb3) CL_ACC(s_poll, sb2, sb3)
#ifdef XPS_CL_STAMP
        sb2 = sb1;
#endif
    };
    {
        // interior: persistent launch over every iteration, same-XCD cluster; the slots s .. s + 1 touch iterations s - 3 .. s + 4,
        // all of them before the last processing step (which has no exchange rows or is the dh0 step)
        const int n_it = p.c_end;
        const bool can_fast = p.handoff && fast && has_c && has_g && p.c_begin == 0 && p.g_begin == 0 && p.g_end == n_it;
        int s = s_lo;
        while (s <= s_hi) {
            if (can_fast && ((s - pair) & 1) != 0 && s - 3 >= 0 && s + 4 < n_it - NR) { slots_fast(s); s += 2; }
            else { slot_general(s); ++s; }
        }
    }
#ifdef XPS_CL_STAMP
    if (lane == 0) { const int wid = (blockIdx.x * 8 + wave) & 2047; g_clstamp[wid * 8 + 0] = s_work; g_clstamp[wid * 8 + 1] = s_drain; g_clstamp[wid * 8 + 2] = s_bar; g_clstamp[wid * 8 + 3] = s_poll; }
#endif
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
struct ClPlan {
    bool ok;
    int KSPLIT, KP, U, CS, nblk, Mc, NR, Bp, grid;
    size_t flags_bytes, xbuf_fwd, xbuf_bwd, keep_bytes;
};

// Per-device facts, looked up for the CURRENT device of the calling thread (one process may drive several GPUs): the CU
// count, how many workgroups of each cluster kernel the device holds at once (occupancy query x CUs: the persistent form
// spins across workgroups, so its whole grid must be co-resident), and the caller's sticky status word.
struct ClDev {
    int cus = -1;
    // [kernel kind: forward / 1-D BPTT / 2-D BPTT][fp32 / bf16x3]; the (kernel, LDS bytes) pair is part of the key: the 2-D BPTT
    // kernel has another LDS block and register count than the 1-D one.  Relaxed atomics: the autograd thread and
    // the main thread may both ask; either computes the same value.
    std::atomic<int> resident[5][2];
    ClDev() { for (auto& k : resident) for (auto& v : k) v.store(-1, std::memory_order_relaxed); }
    unsigned* sticky = nullptr;
};
constexpr int CL_MAX_DEV = 64;
ClDev g_cldev[CL_MAX_DEV];

int cl_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CL_MAX_DEV) return -1;
    return dev;
}

int cl_num_cus() {
    const int dev = cl_device();
    if (dev < 0) return 0;
    if (g_cldev[dev].cus < 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        g_cldev[dev].cus = cus;
    }
    return g_cldev[dev].cus;
}

// mode: 0 = off (per-step GEMM kernels of xps_gru.hip), 1 = cluster kernels one step per launch, 2 = persistent (default)
int g_cluster_mode = -1;
int cl_mode() {
    if (g_cluster_mode < 0) {
        const char* e = getenv("XPS_GRU_CLUSTER");
        int m = 2;
        if (e && (!strcmp(e, "off") || !strcmp(e, "0"))) m = 0;
        else if (e && (!strcmp(e, "steps") || !strcmp(e, "1"))) m = 1;
        g_cluster_mode = m;
    }
    return g_cluster_mode;
}

ClPlan cl_plan(int B, int H, int ndir) {
    ClPlan pl;
    memset(&pl, 0, sizeof(pl));
    if (cl_mode() == 0 || H <= 256 || H > 512 || (H % 4) != 0 || B < 128) return pl;
    int cus = cl_num_cus();
    {   // diagnostic: plan for fewer CUs than the device has (how the launch time scales with the CUs that take part: DESIGN.md 4.5)
        static const int cap = [] { const char* e = getenv("XPS_GRU_CL_CUS"); return e ? atoi(e) : 0; }();
        if (cap > 0 && cap < cus) cus = cap;
    }
    pl.KSPLIT = H > 256 ? 2 : 1;
    pl.KP = 256 * pl.KSPLIT;
    pl.U = 64 / pl.KSPLIT;
    pl.CS = (H + pl.U - 1) / pl.U;
    const int max_blk = cus / (pl.CS * ndir);
    if (max_blk < 1) return pl;
    // >= 6 rounds of 32 trials per cluster: the hand-off pipeline needs 4 (flags lag two rounds, polls lead two), and the gate
    // waves request a round's inputs two rounds early: with 6 the previous step's h of the same trials (stored by the same lane)
    // left the wave at least three rounds before it is requested again.  Small batches get empty (masked) rounds.
    int nblk = B / 192;
    if (nblk > max_blk) nblk = max_blk;
    if (nblk < 1) nblk = 1;
    pl.nblk = nblk;
    const int per = (B + nblk - 1) / nblk;
    pl.Mc = ((per + 31) / 32) * 32;
    if (pl.Mc < 192) pl.Mc = 192;
    pl.NR = pl.Mc / 32;
    pl.Bp = pl.nblk * pl.Mc;
    pl.grid = ndir * pl.nblk * pl.CS;
    // header: flags [cluster][NR][16], XCC table [cluster][16], padding, status block (last 256 B); zeroed before every launch
    pl.flags_bytes = (((size_t)ndir * pl.nblk * (pl.NR + 1) * 16 * 4 + 256 + 255) / 256) * 256;
    pl.xbuf_fwd = (size_t)2 * ndir * pl.Bp * pl.KP * 4;
    pl.xbuf_bwd = 3 * pl.xbuf_fwd;
    pl.keep_bytes = (((size_t)ndir * B * pl.CS * 32 * 4 + 255) / 256) * 256;
    pl.ok = pl.NR >= 6 && pl.xbuf_bwd < ((size_t)1 << 31);
    return pl;
}

// 2-D cluster BPTT (gru_cluster2_bwd_kernel): bf16x3 mode, 384 < H <= 512, clusters of 16 = 4 x 4, >= 16 rounds of 16 trials
// per cluster (the two hand-offs of a step span 14 slots of the round pipeline).
// Opt-in (xps_set_gru_bptt_grid(1) / XPS_GRU_CL2=1).  Measured on the configs[3] shard (one bidirectional H = 512 layer, 2048 trials,
// 20 steps): 1.00-1.03 ms per launch against 0.98-1.02 ms of the 1-D kernel, step 7.45 vs 7.30 ms, HBM-side traffic 4.0 GB vs
// 3.4 GB (PMC): both kernels move their bytes at ~3.5-4 TB/s and NONE of the exchanged bytes is served from the XCD's L2 (they
// are consumed a step after they were written: 25 MB of stream traffic per XCD later), so the second hand-off of the 2-D grid
// (the partial-sum quarters: +1 GB written and re-fetched) costs what its three-fold smaller LDS-DMA ingest saves.  DESIGN.md 4.5.
int g_cl2 = -1;
bool cl2_enabled() {
    if (g_cl2 < 0) { const char* e = getenv("XPS_GRU_CL2"); g_cl2 = (e && e[0] == '1') ? 1 : 0; }
    return g_cl2 != 0;
}
struct ClPlan2 {
    bool ok;
    int nblk, Mc, NR, Bp, grid;
    size_t flags_bytes, keep_bytes, xbuf, pbuf;
};
ClPlan2 cl_plan2(int B, int H, int ndir) {
    ClPlan2 pl;
    memset(&pl, 0, sizeof(pl));
    if (!cl2_enabled() || cl_mode() == 0 || H <= 384 || H > 512 || (H % 4) != 0 || B < 128) return pl;
    const int max_blk = cl_num_cus() / (16 * ndir);
    if (max_blk < 1) return pl;
    int nblk = B / 256;
    if (nblk > max_blk) nblk = max_blk;
    if (nblk < 1) nblk = 1;
    pl.nblk = nblk;
    const int per = (B + nblk - 1) / nblk;
    pl.Mc = ((per + 31) / 32) * 32;
    if (pl.Mc < 256) pl.Mc = 256;
    pl.NR = pl.Mc / C2_RT;
    pl.Bp = pl.nblk * pl.Mc;
    pl.grid = ndir * pl.nblk * 16;
    // header: flags [cluster][NR][160], XCC table [cluster][16], padding, status block (last 256 B); zeroed before every launch
    pl.flags_bytes = (((size_t)ndir * pl.nblk * (pl.NR * 160 + 16) * 4 + 256 + 255) / 256) * 256;
    pl.keep_bytes = (((size_t)ndir * B * 16 * 32 * 4 + 255) / 256) * 256;
    pl.xbuf = (size_t)2 * ndir * (pl.Bp / C2_RT) * 4 * C2_IMG;
    pl.pbuf = (size_t)2 * ndir * pl.nblk * pl.NR * 64 * 2048;
    pl.ok = pl.xbuf < ((size_t)1 << 31) && pl.pbuf < ((size_t)1 << 32);
    return pl;
}

template <typename K>
bool cl_set_lds(K kernel, int bytes) {
    return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}

inline bool cl_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// workgroups of `kernel` (512 threads, `lds` bytes) the current device holds at once; cached per device and kernel kind
// (kind: 0 forward, 1 one-dimensional BPTT, 2 two-dimensional BPTT, 3 forward with the image exchange, 4 one-dimensional BPTT with its
// outputs as exchange rows -- each kind is ONE (kernel, lds) pair per precision)
template <typename K>
int cl_resident(K kernel, int lds, int kind, int bf) {
    const int dev = cl_device();
    if (dev < 0) return 0;
    std::atomic<int>& slot = g_cldev[dev].resident[kind][bf];
    int v = slot.load(std::memory_order_relaxed);
    if (v < 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kernel, 512, (size_t)lds) != hipSuccess) per_cu = 0;
        v = per_cu * cl_num_cus();
        slot.store(v, std::memory_order_relaxed);
    }
    return v;
}

// XPS_GRU_XIMG=0: the forward kernel keeps its ring buffer even when it writes the image of y_ext (A/B; read per call)
bool cl_ximg_enabled() {
    const char* e = getenv("XPS_GRU_XIMG");
    return !(e && e[0] == '0');
}

// XPS_GRU_XOUT=1: the 1-D BPTT kernel reads its exchange rows from its own outputs (A/B; read per call)
bool cl_xout_enabled() {
    const char* e = getenv("XPS_GRU_XOUT");
    return e && e[0] == '1';
}

unsigned* cl_sticky() {
    const int dev = cl_device();
    return dev < 0 ? nullptr : g_cldev[dev].sticky;
}

}  // namespace

// A device word that outlives every workspace: a hand-off that gave up stores 1 there as well (NULL: none).  Per device;
// the caller owns the memory, zeroes it, and reads it where it synchronises anyway.
extern "C" int xps_gru_set_status_word(unsigned* device_word) {
    const int dev = cl_device();
    if (dev < 0) { xps_set_error("xps_gru_set_status_word: no current device"); return XPS_E_HIP; }
    g_cldev[dev].sticky = device_word;
    return XPS_OK;
}

// ---- internal interface used by xps_gru.hip's entry points ----
bool xps_internal_gru_cluster_usable(int B, int H, int ndir) { return cl_plan(B, H, ndir).ok; }

size_t xps_internal_gru_cluster_fwd_workspace(int B, int H, int ndir) {
    const ClPlan pl = cl_plan(B, H, ndir);
    return pl.ok ? pl.flags_bytes + pl.xbuf_fwd : 0;
}

size_t xps_internal_gru_cluster_bwd_workspace(int B, int H, int ndir) {
    const ClPlan pl = cl_plan(B, H, ndir);
    if (!pl.ok) return 0;
    size_t n = pl.flags_bytes + pl.keep_bytes + pl.xbuf_bwd;
    // (both BPTT kernels' needs, whatever the precision mode is when the launch comes: the mode is a run-time switch)
    const ClPlan2 p2 = cl_plan2(B, H, ndir);
    if (p2.ok) { const size_t n2 = p2.flags_bytes + p2.keep_bytes + p2.xbuf + p2.pbuf; if (n2 > n) n = n2; }
    return n;
}

size_t xps_internal_gru_cluster_status_offset(int B, int H, int ndir) {
    const ClPlan pl = cl_plan(B, H, ndir);
    return pl.ok ? pl.flags_bytes - 256 : 0;
}

// the forward launch can use the XPS_FMT_SPLIT4 image of y_ext as its exchange buffer (bf16x3 mode is the caller's check)
bool xps_internal_gru_cluster_ximg_ok(int B, int H, int ndir) {
    const ClPlan pl = cl_plan(B, H, ndir);
    return pl.ok && H == pl.KP && pl.Bp == B && cl_ximg_enabled();
}

int xps_internal_gru_cluster_fwd(const float* gi, const float* const* w_hh, const float* const* b_hh, const float* h0,
                                 float* y_ext, float* saved, int T, int B, int H, int ndir, void* workspace, hipStream_t st,
                                 float* y_split, float* yd_split, float drop_p, unsigned long long drop_seed) {
    const ClPlan pl = cl_plan(B, H, ndir);
    if (!pl.ok) { xps_set_error("gru cluster forward: unsupported shape"); return XPS_E_INVALID; }
    const bool bf = xps_internal_gemm_mode() == 1;
    if ((y_split || yd_split) && (!bf || H % 4 != 0 || !cl_aligned16(y_split) || !cl_aligned16(yd_split))) {
        xps_set_error("gru cluster forward: XPS_FMT_SPLIT4 images exist in bf16x3 mode only (H a multiple of 4, 16-byte aligned)");
        return XPS_E_INVALID;
    }
    ClFwd p;
    p.gi = gi; p.y_ext = y_ext; p.saved = saved;
    p.y_split = y_split; p.yd_split = yd_split; p.drop_p = drop_p; p.drop_scale = 1.0f / (1.0f - drop_p); p.drop_seed = drop_seed;
    // the image of y_ext as the exchange buffer itself (gru_cluster_fwd_kernel<.., XIMG>): rows of exactly KP elements, no pad trials
    const bool ximg = bf && y_split && H == pl.KP && pl.Bp == B && cl_ximg_enabled();
    for (int d = 0; d < 2; ++d) { p.w_hh[d] = w_hh[d < ndir ? d : 0]; p.b_hh[d] = b_hh[d < ndir ? d : 0]; }
    if (!cl_aligned16(gi) || !cl_aligned16(y_ext) || !cl_aligned16(saved) || !cl_aligned16(p.w_hh[0]) || !cl_aligned16(p.w_hh[1]) ||
        !cl_aligned16(p.b_hh[0]) || !cl_aligned16(p.b_hh[1]) || !cl_aligned16(h0) || !cl_aligned16(workspace)) {
        xps_set_error("gru cluster forward: operands must be 16-byte aligned");
        return XPS_E_INVALID;
    }
    unsigned char* ws = (unsigned char*)workspace;
    p.flags = (unsigned*)ws;
    p.xcc = p.flags + (size_t)ndir * pl.nblk * pl.NR * 16;
    p.status = (unsigned*)(ws + pl.flags_bytes - 256);
    p.xbuf = ws + pl.flags_bytes;
    p.xbuf_bytes = (unsigned)pl.xbuf_fwd;
    p.T = T; p.B = B; p.H = H; p.ndir = ndir; p.Bp = pl.Bp; p.Mc = pl.Mc; p.NR = pl.NR; p.nblk = pl.nblk; p.CS = pl.CS;
    p.saved_mm = (H % 32 == 0) ? 1 : 0;
    if (pl.CS * pl.U < pl.KP) {
        // state columns no member owns (H far below KP) meet zero weights in the contraction: they must be finite
        if (hipMemsetAsync(p.xbuf, 0, pl.xbuf_fwd, st) != hipSuccess) { xps_set_error("gru cluster forward: memset failed"); return XPS_E_HIP; }
    }
    {
        const long long total = (long long)ndir * pl.Bp * (pl.KP / 4);
        const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
        // (flags_bytes is a multiple of 256; the workspace is 16-byte aligned: checked above)
        if (bf) hipLaunchKernelGGL(gru_cluster_init_kernel<true>, dim3(blocks), dim3(256), 0, st, h0, y_ext, ximg ? (void*)nullptr : p.xbuf, T, B, H, ndir, pl.Bp, pl.KP, (u32x4*)ws, (int)(pl.flags_bytes / 16), y_split);
        else hipLaunchKernelGGL(gru_cluster_init_kernel<false>, dim3(blocks), dim3(256), 0, st, h0, y_ext, p.xbuf, T, B, H, ndir, pl.Bp, pl.KP, (u32x4*)ws, (int)(pl.flags_bytes / 16), (float*)nullptr);
    }
    p.sticky = cl_sticky();
    auto launch = [&](auto kernel, int lds) -> bool {
        if (!cl_set_lds(kernel, lds)) return false;
        // the persistent form spins across workgroups: only when the device holds the whole grid at once
        const bool persistent = cl_mode() == 2 && pl.grid <= cl_resident(kernel, lds, ximg ? 3 : 0, bf ? 1 : 0);
        if (persistent) {
            p.s_begin = 0; p.s_end = T; p.handoff = 1;
            hipLaunchKernelGGL(kernel, dim3(pl.grid), dim3(512), lds, st, p);
        } else {
            p.handoff = 0;
            for (int s = 0; s < T; ++s) {
                p.s_begin = s; p.s_end = s + 1;
                hipLaunchKernelGGL(kernel, dim3(pl.grid), dim3(512), lds, st, p);
            }
        }
        return true;
    };
    bool ok;
    ok = ximg ? launch(gru_cluster_fwd_kernel<2, true, true>, ClCfg<2, true, 1>::LDS_BYTES)
       : bf ? launch(gru_cluster_fwd_kernel<2, true>, ClCfg<2, true, 1>::LDS_BYTES) : launch(gru_cluster_fwd_kernel<2, false>, ClCfg<2, false, 1>::LDS_BYTES);
    if (!ok) { xps_set_error("gru cluster forward: cannot raise the dynamic LDS limit"); return XPS_E_HIP; }
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

int xps_internal_gru_cluster_bwd(const float* dy, const float* dhn, const float* y_ext, const float* saved,
                                 const float* const* w_hh_t, float* dgi, float* dghn, float* dh0,
                                 int T, int B, int H, int ndir, void* workspace, hipStream_t st, int split_out) {
    const ClPlan pl = cl_plan(B, H, ndir);
    if (!pl.ok) { xps_set_error("gru cluster backward: unsupported shape"); return XPS_E_INVALID; }
    const bool bf = xps_internal_gemm_mode() == 1;
    if (split_out && !bf) { xps_set_error("gru cluster backward: XPS_FMT_SPLIT4 outputs exist in bf16x3 mode only"); return XPS_E_INVALID; }
    const ClPlan2 p2 = cl_plan2(B, H, ndir);
    if (bf && p2.ok) {
        ClBwd2 q;
        q.dy = dy; q.dhn = dhn; q.y_ext = y_ext; q.saved = saved; q.dgi = dgi; q.dghn = dghn; q.dh0 = dh0;
        q.split_out = split_out;
        for (int d = 0; d < 2; ++d) q.w_hh_t[d] = w_hh_t[d < ndir ? d : 0];
        if (!cl_aligned16(dy) || !cl_aligned16(dhn) || !cl_aligned16(y_ext) || !cl_aligned16(saved) || !cl_aligned16(q.w_hh_t[0]) ||
            !cl_aligned16(q.w_hh_t[1]) || !cl_aligned16(dgi) || !cl_aligned16(dghn) || !cl_aligned16(dh0) || !cl_aligned16(workspace)) {
            xps_set_error("gru cluster backward: operands must be 16-byte aligned");
            return XPS_E_INVALID;
        }
        unsigned char* ws = (unsigned char*)workspace;
        q.flags = (unsigned*)ws;
        q.xcc = q.flags + (size_t)ndir * p2.nblk * p2.NR * 160;
        q.status = (unsigned*)(ws + p2.flags_bytes - 256);
        q.sticky = cl_sticky();
        q.keep = (float*)(ws + p2.flags_bytes);
        q.xbuf = ws + p2.flags_bytes + p2.keep_bytes;
        q.pbuf = (float*)(ws + p2.flags_bytes + p2.keep_bytes + p2.xbuf);
        q.xbuf_bytes = (unsigned)p2.xbuf;
        q.pbuf_bytes = (unsigned)p2.pbuf;
        q.T = T; q.B = B; q.H = H; q.ndir = ndir; q.Bp = p2.Bp; q.Mc = p2.Mc; q.NR = p2.NR; q.nblk = p2.nblk;
        q.saved_mm = (H % 32 == 0) ? 1 : 0;
        if (hipMemsetAsync(ws, 0, p2.flags_bytes, st) != hipSuccess) { xps_set_error("gru cluster backward: memset failed"); return XPS_E_HIP; }
        const int ps_total = T + (dh0 ? 1 : 0);
        q.ps_total = ps_total;
        if (!cl_set_lds(gru_cluster2_bwd_kernel, C2_LDS)) { xps_set_error("gru cluster backward: cannot raise the dynamic LDS limit"); return XPS_E_HIP; }
        const int n_it = (ps_total - 1) * p2.NR;
        const bool persistent = cl_mode() == 2 && p2.grid <= cl_resident(gru_cluster2_bwd_kernel, C2_LDS, 2, 1);
        if (persistent) {
            q.handoff = 1; q.do_ps0 = 1; q.c_begin = 0; q.c_end = n_it; q.g_begin = 0; q.g_end = n_it;
            hipLaunchKernelGGL(gru_cluster2_bwd_kernel, dim3(p2.grid), dim3(512), C2_LDS, st, q);
        } else {
            // one step per launch: the gate pass of step 0, then per step a contraction launch and a gate launch
            q.handoff = 0; q.do_ps0 = 1; q.c_begin = q.c_end = q.g_begin = q.g_end = 0;
            hipLaunchKernelGGL(gru_cluster2_bwd_kernel, dim3(p2.grid), dim3(512), C2_LDS, st, q);
            q.do_ps0 = 0;
            for (int ps = 1; ps < ps_total; ++ps) {
                q.c_begin = (ps - 1) * p2.NR; q.c_end = ps * p2.NR; q.g_begin = q.g_end = 0;
                hipLaunchKernelGGL(gru_cluster2_bwd_kernel, dim3(p2.grid), dim3(512), C2_LDS, st, q);
                q.g_begin = q.c_begin; q.g_end = q.c_end; q.c_begin = q.c_end = 0;
                hipLaunchKernelGGL(gru_cluster2_bwd_kernel, dim3(p2.grid), dim3(512), C2_LDS, st, q);
            }
        }
        XPS_CHECK_LAUNCH();
        return XPS_OK;
    }
    ClBwd p;
    p.dy = dy; p.dhn = dhn; p.y_ext = y_ext; p.saved = saved; p.dgi = dgi; p.dghn = dghn; p.dh0 = dh0;
    if (split_out && !bf) { xps_set_error("gru cluster backward: XPS_FMT_SPLIT4 outputs exist in bf16x3 mode only"); return XPS_E_INVALID; }
    p.split_out = split_out;
    for (int d = 0; d < 2; ++d) p.w_hh_t[d] = w_hh_t[d < ndir ? d : 0];
    if (!cl_aligned16(dy) || !cl_aligned16(dhn) || !cl_aligned16(y_ext) || !cl_aligned16(saved) || !cl_aligned16(p.w_hh_t[0]) ||
        !cl_aligned16(p.w_hh_t[1]) || !cl_aligned16(dgi) || !cl_aligned16(dghn) || !cl_aligned16(dh0) || !cl_aligned16(workspace)) {
        xps_set_error("gru cluster backward: operands must be 16-byte aligned");
        return XPS_E_INVALID;
    }
    unsigned char* ws = (unsigned char*)workspace;
    p.flags = (unsigned*)ws;
    p.xcc = p.flags + (size_t)ndir * pl.nblk * pl.NR * 16;
    p.status = (unsigned*)(ws + pl.flags_bytes - 256);
    p.keep = (float*)(ws + pl.flags_bytes);
    p.xbuf = ws + pl.flags_bytes + pl.keep_bytes;
    p.xbuf_bytes = (unsigned)pl.xbuf_bwd;
    p.T = T; p.B = B; p.H = H; p.ndir = ndir; p.Bp = pl.Bp; p.Mc = pl.Mc; p.NR = pl.NR; p.nblk = pl.nblk; p.CS = pl.CS;
    p.saved_mm = (H % 32 == 0) ? 1 : 0;
    // outputs as exchange rows (gru_cluster_bwd_kernel<.., XOUT>): split4 outputs, rows of exactly KP elements, no pad trials
    const bool xout = bf && split_out && H == pl.KP && pl.Bp == B && cl_xout_enabled();
    if (hipMemsetAsync(ws, 0, pl.flags_bytes, st) != hipSuccess) { xps_set_error("gru cluster backward: memset failed"); return XPS_E_HIP; }
    if (pl.CS * pl.U < pl.KP) {
        // gate-gradient columns no member owns (H far below KP) are contracted with zero weights: they must be finite
        if (hipMemsetAsync(p.xbuf, 0, pl.xbuf_bwd, st) != hipSuccess) { xps_set_error("gru cluster backward: memset failed"); return XPS_E_HIP; }
    }
    const int ps_total = T + (dh0 ? 1 : 0);
    p.ps_total = ps_total;
    p.sticky = cl_sticky();
    auto launch = [&](auto kernel, int lds) -> bool {
        if (!cl_set_lds(kernel, lds)) return false;
        const bool persistent = cl_mode() == 2 && pl.grid <= cl_resident(kernel, lds, xout ? 4 : 1, bf ? 1 : 0);
        if (persistent) {
            p.ps_begin = 0; p.ps_end = ps_total; p.handoff = 1;
            hipLaunchKernelGGL(kernel, dim3(pl.grid), dim3(512), lds, st, p);
        } else {
            p.handoff = 0;
            for (int ps = 0; ps < ps_total; ++ps) {
                p.ps_begin = ps; p.ps_end = ps + 1;
                hipLaunchKernelGGL(kernel, dim3(pl.grid), dim3(512), lds, st, p);
            }
        }
        return true;
    };
    bool ok;
    ok = xout ? launch(gru_cluster_bwd_kernel<2, true, true>, ClCfg<2, true, 3>::LDS_BYTES)
       : bf ? launch(gru_cluster_bwd_kernel<2, true>, ClCfg<2, true, 3>::LDS_BYTES) : launch(gru_cluster_bwd_kernel<2, false>, ClCfg<2, false, 3>::LDS_BYTES);
    if (!ok) { xps_set_error("gru cluster backward: cannot raise the dynamic LDS limit"); return XPS_E_HIP; }
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}

// diagnostics (tools/): byte offset of the status block in the BPTT workspace of the kernel the current mode selects
extern "C" long long xps_debug_gru_bwd_status_offset(int B, int H, int ndir) {
    const ClPlan2 p2 = cl_plan2(B, H, ndir);
    if (xps_internal_gemm_mode() == 1 && p2.ok) return (long long)p2.flags_bytes - 256;
    const ClPlan pl = cl_plan(B, H, ndir);
    return pl.ok ? (long long)pl.flags_bytes - 256 : -1;
}

#ifdef XPS_CL_STAMP
extern "C" int xps_debug_read_cluster_stamps2(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_clstamp2), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -2;
}
extern "C" int xps_debug_read_cluster_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_clstamp), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int xps_set_gru_cluster_mode(int mode) {
    if (mode < 0 || mode > 2) { xps_set_error("xps_set_gru_cluster_mode: mode must be 0 (off), 1 (one step per launch) or 2 (persistent)"); return XPS_E_INVALID; }
    g_cluster_mode = mode;
    return XPS_OK;
}
extern "C" int xps_get_gru_cluster_mode(void) { return cl_mode(); }
extern "C" int xps_set_gru_bptt_grid(int two_dimensional) {
    if (two_dimensional != 0 && two_dimensional != 1) { xps_set_error("xps_set_gru_bptt_grid: 0 (1-D cluster kernel) or 1 (4 x 4 grid)"); return XPS_E_INVALID; }
    g_cl2 = two_dimensional;
    return XPS_OK;
}
extern "C" int xps_get_gru_bptt_grid(void) { return cl2_enabled() ? 1 : 0; }
