// Histograms behind the min-sum normalisation fit of rework/Alvarado.py:10-66.
//
// The reference collects the first-iteration check->variable messages of `trials` decodes, splits
// them by the true value of the bit they talk about (:33-36), bins both sets with np.histogram over
// their common range (:44-49) and fits log(hist_0 / hist_1) = alpha * lambda (:51-62).  Here the
// messages never leave the device: the general-H kernel dumps them (qbp_generic.hpp, dump_R), one
// kernel finds their range, a second one bins them by class with LDS atomics.  Bin edges are
// np.linspace(lo, hi, bins + 1) computed on the host the way numpy does (start + i * step, last edge
// = hi), and a value's bin is the one whose edges contain it (left-closed, the last bin closed on
// both sides) -- numpy's own rule after its rounding corrections, so the counts are numpy's.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qbp {

#ifdef QBP_DEFINE_KERNELS   /* non-template kernels: defined in their translation unit only */
// per-block minimum / maximum of x[0 .. count)
__global__ __launch_bounds__(256) void hist_minmax_kernel(const double* x, long long count, double* part /*[2 * grid]*/)
{
    __shared__ double smin[256], smax[256];
    double lo = __builtin_inf(), hi = -__builtin_inf();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        const double v = x[i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smin[threadIdx.x] = smin[threadIdx.x + s] < smin[threadIdx.x] ? smin[threadIdx.x + s] : smin[threadIdx.x];
            smax[threadIdx.x] = smax[threadIdx.x + s] > smax[threadIdx.x] ? smax[threadIdx.x + s] : smax[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = smin[0]; part[2 * blockIdx.x + 1] = smax[0]; }
}

// messages [B][E] (CSR edge order), errors [B][n]: class of message (b, e) = errors[b][col_idx[e]].
// hist [2][bins] (u64, added to).  edges [bins + 1].
__global__ __launch_bounds__(256) void hist_bin_kernel(const double* msg, const uint8_t* errors,
                                                       const int32_t* col_idx, long long B, int E, int n,
                                                       const double* edges, int bins,
                                                       unsigned long long* hist)
{
    extern __shared__ unsigned lh[];                 // [2][bins] counts, then [bins + 1] edges as doubles
    double* const le = reinterpret_cast<double*>(lh + ((2 * bins + 1) & ~1));
    for (int i = threadIdx.x; i < 2 * bins; i += blockDim.x) lh[i] = 0u;
    for (int i = threadIdx.x; i <= bins; i += blockDim.x) le[i] = edges[i];
    __syncthreads();
    const double first = le[0], last = le[bins];
    const double norm = (double)bins / (last - first);
    const long long total = B * (long long)E;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / E;
        const int e = (int)(i - b * E);
        const double v = msg[i];
        if (!(v >= first && v <= last)) continue;      // (NaN or outside: np.histogram drops them)
        int k = (int)((v - first) * norm);
        k = k < 0 ? 0 : (k > bins - 1 ? bins - 1 : k);
        while (k > 0 && v < le[k]) --k;                 // the estimate can be one off next to an edge
        while (k < bins - 1 && v >= le[k + 1]) ++k;
        const unsigned cls = errors[b * n + col_idx[e]] & 1u;
        atomicAdd(&lh[cls * bins + k], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * bins; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

#endif  // QBP_DEFINE_KERNELS

}  // namespace qbp
