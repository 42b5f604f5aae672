// Per-bin high-gamma feature extraction of the realtime pipeline (realtime_sim/realtime_processing.py:10-164
// process_HG): common average reference -> one IIR/FIR band-pass per band with carried filter state -> RMS over
// (time, bands) per channel.  One launch per 20 ms bin; float64 like the reference (numpy / scipy.signal.lfilter).
//
// A bin is tiny (128 channels x 40 samples x ~8 bands): the kernel is latency-bound; what it buys is that the features
// are produced on the device, in one launch, next to the hipGraph-captured GRU step that consumes them.
//
// Arithmetic order follows the reference bit for bit where it is defined:
//  * CAR: np.mean(data[good], axis=0) = sequential sum over the good channels in index order, / count; data - avg
//  * lfilter: scipy's direct-form-II-transposed loop, coefficients normalised by a[0], NO fused multiply-add
//  * power: np.mean(np.square(y), axis=(1, 2)) over the contiguous (time, band) block of a channel = numpy's pairwise
//    summation (8 accumulators up to 128 elements, recursive halving above), / count, sqrt
#include "xps_common.h"

namespace {
constexpr int HG_MAXT = 2048, HG_MAXTAPS = 32;

// numpy's pairwise_sum (loops_utils.h.src) on a contiguous array
__device__ double np_pairwise_sum(const double* a, int n) {
#pragma clang fp contract(off)
    // explicit stack of (start, length) segments; the combine order of the recursion is res(left) + res(right), which an
    // in-order traversal with a value stack reproduces
    int seg_start[24], seg_len[24], seg_state[24];
    double val[24];
    int sp = 0, vp = 0;
    seg_start[0] = 0; seg_len[0] = n; seg_state[0] = 0; sp = 1;
    while (sp > 0) {
        const int s = seg_start[sp - 1], len = seg_len[sp - 1], st = seg_state[sp - 1];
        if (len <= 128) {
            double res;
            if (len < 8) {
                res = 0.0;
                for (int i = 0; i < len; ++i) res += a[s + i];
            } else {
                double r[8];
                for (int j = 0; j < 8; ++j) r[j] = a[s + j];
                int i;
                for (i = 8; i < len - (len % 8); i += 8)
                    for (int j = 0; j < 8; ++j) r[j] += a[s + i + j];
                res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (; i < len; ++i) res += a[s + i];
            }
            val[vp++] = res;
            --sp;
        } else if (st == 0) {                       // descend into the left half
            int n2 = len / 2;
            n2 -= n2 % 8;
            seg_state[sp - 1] = 1;
            seg_start[sp] = s; seg_len[sp] = n2; seg_state[sp] = 0; ++sp;
        } else if (st == 1) {                       // then the right half
            int n2 = len / 2;
            n2 -= n2 % 8;
            seg_state[sp - 1] = 2;
            seg_start[sp] = s + n2; seg_len[sp] = len - n2; seg_state[sp] = 0; ++sp;
        } else {                                    // combine
            const double right = val[--vp], left = val[--vp];
            val[vp++] = left + right;
            --sp;
        }
    }
    return val[0];
}

// one block = HG_CPB channels; thread -> (local channel, band)
constexpr int HG_CPB = 8;
__global__ void process_hg_kernel(const double* __restrict__ data, int C, int Tn, const unsigned char* __restrict__ good,
                                  const double* __restrict__ bcoef, const double* __restrict__ acoef, int bands, int taps,
                                  double* __restrict__ zi, int do_car, double* __restrict__ car_out,
                                  double* __restrict__ filtered, double* __restrict__ power) {
#pragma clang fp contract(off)
    __shared__ double avg[HG_MAXT];
    const int tid = threadIdx.x, nthr = blockDim.x;
    // 1. common average over the good channels (every block recomputes it: C x Tn doubles, L2-resident)
    if (do_car) {
        int ngood = 0;
        for (int c = 0; c < C; ++c) ngood += (!good || good[c]) ? 1 : 0;
        for (int t = tid; t < Tn; t += nthr) {
            double s = 0.0;
            for (int c = 0; c < C; ++c)
                if (!good || good[c]) s += data[(long long)c * Tn + t];
            avg[t] = s / (double)ngood;
        }
    } else {
        for (int t = tid; t < Tn; t += nthr) avg[t] = 0.0;
    }
    __syncthreads();
    const int c0 = blockIdx.x * HG_CPB;
    if (car_out)
        for (int i = tid; i < HG_CPB * Tn; i += nthr) {
            const int c = c0 + i / Tn, t = i % Tn;
            if (c < C) car_out[(long long)c * Tn + t] = do_car ? data[(long long)c * Tn + t] - avg[t] : data[(long long)c * Tn + t];
        }
    // 2. filters: thread (cl, band)
    const int cl = tid / bands, band = tid % bands;
    const int c = c0 + cl;
    if (bands > 0 && cl < HG_CPB && c < C) {
        double b[HG_MAXTAPS], a[HG_MAXTAPS], z[HG_MAXTAPS];
        const double a0 = acoef ? acoef[(long long)band * taps] : 1.0;
        for (int k = 0; k < taps; ++k) {
            b[k] = bcoef[(long long)band * taps + k] / a0;
            a[k] = acoef ? acoef[(long long)band * taps + k] / a0 : (k == 0 ? 1.0 : 0.0);
        }
        double* zp = zi ? zi + ((long long)band * C + c) * (taps - 1) : nullptr;
        for (int k = 0; k < taps - 1; ++k) z[k] = zp ? zp[k] : 0.0;
        const double* xr = data + (long long)c * Tn;
        double* yo = filtered + ((long long)c * Tn) * bands + band;
        for (int t = 0; t < Tn; ++t) {
            const double x = do_car ? xr[t] - avg[t] : xr[t];
            double y;
            if (taps > 1) {
                y = z[0] + b[0] * x;
                for (int k = 0; k < taps - 2; ++k) z[k] = z[k + 1] + x * b[k + 1] - y * a[k + 1];
                z[taps - 2] = x * b[taps - 1] - y * a[taps - 1];
            } else {
                y = x * b[0];
            }
            yo[(long long)t * bands] = y;
        }
        if (zp)
            for (int k = 0; k < taps - 1; ++k) zp[k] = z[k];
    }
    __syncthreads();
    // 3. RMS over the contiguous (time, band) block of each channel, numpy's summation order
    if (power && bands > 0 && tid < HG_CPB && c0 + tid < C) {
        double* blk = filtered + (long long)(c0 + tid) * Tn * bands;
        const int n = Tn * bands;
        for (int i = 0; i < n; ++i) blk[i] = blk[i] * blk[i];          // np.square (in place on the scratch copy)
        power[c0 + tid] = sqrt(np_pairwise_sum(blk, n) / (double)n);
    }
}
}  // namespace

extern "C" size_t xps_process_hg_f64_workspace(int C, int Tn, int bands) {
    if (C < 1 || Tn < 1 || bands < 1) return 16;
    return (size_t)C * Tn * bands * sizeof(double) + 16;
}

extern "C" int xps_process_hg_f64(const double* data, int C, int Tn, const uint8_t* good, const double* b, const double* a,
                                  int bands, int taps, double* zi, int do_car, double* car_out, double* filtered,
                                  double* power, void* workspace, size_t workspace_bytes, void* stream) {
    XPS_CHECK_ARG(data && C >= 1 && Tn >= 1 && bands >= 0, "bad argument");
    XPS_CHECK_ARG(Tn <= HG_MAXT, "bin longer than 2048 samples");
    XPS_CHECK_ARG(bands == 0 || (b && taps >= 1 && taps <= HG_MAXTAPS), "1..32 filter taps");
    XPS_CHECK_ARG(bands <= 32, "at most 32 bands");
    XPS_CHECK_ARG(!power || bands > 0, "band power needs at least one band");
    double* filt = filtered;
    if (bands > 0 && !filt) {                     // the squared copy lives in the workspace when the caller does not want y
        if (!workspace || workspace_bytes < xps_process_hg_f64_workspace(C, Tn, bands)) {
            xps_set_error("xps_process_hg_f64: workspace too small");
            return XPS_E_WORKSPACE;
        }
        filt = (double*)workspace;
    }
    XPS_CHECK_ARG(!(filtered && power), "ask for the filtered signal OR the band power in one call (the power pass squares in place)");
    const int threads = bands > 0 ? ((HG_CPB * bands + 63) / 64) * 64 : 64;
    hipLaunchKernelGGL(process_hg_kernel, dim3(cdiv(C, HG_CPB)), dim3(threads), 0, (hipStream_t)stream, data, C, Tn,
                       (const unsigned char*)good, b, a, bands, taps, zi, do_car, car_out, filt, power);
    XPS_CHECK_LAUNCH();
    return XPS_OK;
}
