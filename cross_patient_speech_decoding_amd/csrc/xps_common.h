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
    __host__ __device__ inline long long off(int i) const {
        return (long long)(i / rpg) * gs + (long long)(i % rpg) * ld;
    }
};

static inline RowMap to_rowmap(const xps_rowmap* r) {
    RowMap m;
    m.gs = r->gs;
    m.ld = r->ld;
    m.rpg = r->rpg < 1 ? 1 : r->rpg;
    return m;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

__device__ inline float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

// product precision of the matrix kernels (xps_set_gemm_precision): 0 = fp32 MFMA, 1 = bf16 split products
int xps_internal_gemm_mode();
