// Shared helpers for the gfx950 kernels of libxps.so.  CDNA4 only: wave = 64 lanes,
// f32/f64-input MFMA, 160 KiB LDS per CU.  No CUDA / multi-backend paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/xps.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

void xps_set_error(const char* fmt, ...);

#define XPS_CHECK_ARG(cond, msg)                                  \
    do {                                                          \
        if (!(cond)) {                                            \
            xps_set_error("%s: %s", __func__, msg);               \
            return XPS_E_INVALID;                                 \
        }                                                         \
    } while (0)

#define XPS_CHECK_LAUNCH()                                                        \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            xps_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return XPS_E_HIP;                                                     \
        }                                                                         \
    } while (0)

struct RowMap {
    long long gs, ld;
    int rpg;
    int fmt = 0;      // XPS_FMT_F32 / XPS_FMT_SPLIT4 (xps.h): element format of the operand the map addresses
    __host__ __device__ inline long long off(int i) const {
        return (long long)(i / rpg) * gs + (long long)(i % rpg) * ld;
    }
};

static inline RowMap to_rowmap(const xps_rowmap* r) {
    RowMap m;
    m.gs = r->gs;
    m.ld = r->ld;
    m.rpg = r->rpg < 1 ? 1 : r->rpg;
    m.fmt = r->fmt;
    return m;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ inline float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// counter-based RNG (splitmix64 finaliser over seed + element-pair index): two 24-bit uniforms per hash
__device__ inline void rng_pair(unsigned long long seed, long long pair, float& u0, float& u1) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(pair + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    u0 = (float)((unsigned)(z >> 40)) * (1.0f / 16777216.0f);
    u1 = (float)((unsigned)(z >> 8) & 0xFFFFFFu) * (1.0f / 16777216.0f);
}
// keep decisions (1 / 0) of the four elements 4 q .. 4 q + 3 of a dropout site: THE definition every kernel uses
// (xps_dropout_f32 and the GRU kernels that fuse the inter-layer dropout must agree bit for bit)
// XPS_FMT_SPLIT4 (xps.h): the four fp32 values of an aligned group as bf16 hi[0..3] | lo[0..3] in the same 16 bytes,
// hi = bf16(x), lo = bf16(x - hi): the split the bf16x3 tile kernels make while staging (xps_gemm_tile.h: bf_split).
__device__ inline f32x4 split4_pack(const f32x4 v) {
    typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
    bf16x8_ o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 h = (__bf16)v[j];
        o[j] = h;
        o[4 + j] = (__bf16)(v[j] - (float)h);
    }
    return __builtin_bit_cast(f32x4, o);
}
__device__ inline f32x4 dropout_keep4(unsigned long long seed, long long q, float p) {
    float u0, u1, u2, u3;
    rng_pair(seed, 2 * q, u0, u1);
    rng_pair(seed, 2 * q + 1, u2, u3);
    return (f32x4){u0 >= p ? 1.f : 0.f, u1 >= p ? 1.f : 0.f, u2 >= p ? 1.f : 0.f, u3 >= p ? 1.f : 0.f};
}

// product precision of the matrix kernels (xps_set_gemm_precision): 0 = fp32 MFMA, 1 = bf16 split products
int xps_internal_gemm_mode();
