// Error reporting and ABI version of libxps.so.
#include <stdarg.h>
#include "xps_common.h"

static thread_local char g_err[512] = "";

void xps_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* xps_last_error(void) { return g_err; }
// 2: round 2 (xps_gru_seq_fwd_f32 takes a workspace; fused-dropout, decoder-select, big-tile and augmentation entry points added)
extern "C" int xps_abi_version(void) { return 4; }

// A stream at the LOWEST priority the device offers (torch exposes only "default" and "high"): the side stream
// of the weight-gradient GEMMs must never be preferred over the critical path when a CU frees up.
extern "C" int xps_stream_create_low_priority(void** stream) {
    if (!stream) { xps_set_error("xps_stream_create_low_priority: null argument"); return XPS_E_INVALID; }
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = 0; }
    hipStream_t s = nullptr;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) != hipSuccess) {
        xps_set_error("xps_stream_create_low_priority: hipStreamCreateWithPriority failed");
        return XPS_E_HIP;
    }
    *stream = (void*)s;
    return XPS_OK;
}

extern "C" int xps_stream_destroy(void* stream) {
    if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) return XPS_E_HIP;
    return XPS_OK;
}
