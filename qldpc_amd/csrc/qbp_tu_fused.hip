// libqbp.so, translation unit of the on-chip kernel (qbp_kernels.hpp): its 27 instantiations.
// Compiled twice: as is (numpy's tanh / arctanh: the default), and with -DQBP_FAST_TU -DQBP_MATH_FAST=1 under
// other names (QBP_FLAG_FAST_MATH: round 2's approximations); each build is a code object of its own.
#ifdef QBP_FAST_TU
#define bp_fused_kernel bp_fused_kernel_fast_math
#define launch_fused launch_fused_fast_math
#else
#define QBP_DEFINE_KERNELS 1
#endif
#include <hip/hip_runtime.h>

#include "../../include/qbp.h"
#include "qbp_kernels.hpp"
#include "qbp_launch.hpp"

namespace qbp {
namespace {

template <int DC, int DV, int VARIANT, bool MC, bool FORCE, int MAXT, int MINW, bool ONEBAR = false>
hipError_t launch_k(const FusedParams& P, const LaunchCfg& cfg, hipStream_t stream)
{
    auto kern = bp_fused_kernel<DC, DV, VARIANT, MC, FORCE, MAXT, MINW, ONEBAR>;
    // the dynamic-LDS limit of an instantiation is raised once per device and size (the attribute
    // call costs a few microseconds, which matters for one-syndrome-per-call users)
    static thread_local int lds_set[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || lds_set[dev] < cfg.lds_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, cfg.lds_bytes);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) lds_set[dev] = cfg.lds_bytes;
    }
    hipLaunchKernelGGL(kern, dim3(cfg.grid), dim3(cfg.threads), cfg.lds_bytes, stream, P);
    return hipGetLastError();
}

template <int VARIANT, bool MC, int MAXT, int MINW = 1>
hipError_t launch_one(const FusedParams& P, const LaunchCfg& cfg, hipStream_t stream)
{
    const bool force = (P.flags & QBP_FLAG_FORCE_FULL) != 0;
    if constexpr (!MC) {      // (the wide shape is 1.5 % slower with one barrier -- more spills -- and keeps two)
        if (force && cfg.one_barrier && cfg.dc == DC_SMALL)
            return launch_k<DC_SMALL, DV_SMALL, VARIANT, false, true, MAXT, MINW, true>(P, cfg, stream);
    }
    if (cfg.dc == DC_SMALL)
        return force ? launch_k<DC_SMALL, DV_SMALL, VARIANT, MC, true, MAXT, MINW>(P, cfg, stream)
                     : launch_k<DC_SMALL, DV_SMALL, VARIANT, MC, false, MAXT, MINW>(P, cfg, stream);
    return force ? launch_k<DC_WIDE, DV_WIDE, VARIANT, MC, true, MAXT, MINW>(P, cfg, stream)
                 : launch_k<DC_WIDE, DV_WIDE, VARIANT, MC, false, MAXT, MINW>(P, cfg, stream);
}

template <bool MC>
hipError_t launch_variant(int variant, const FusedParams& P, const LaunchCfg& cfg, hipStream_t s)
{
    // __launch_bounds__(1024): 128-VGPR budget = 4 wavefronts per SIMD, the fastest geometry
    // measured (profiles/r01_tune.txt; builds with 3 or 5 waves per SIMD were slower).
    switch (variant) {
        case QBP_SUM_PRODUCT: return launch_one<0, MC, FUSED_MAX_THREADS>(P, cfg, s);
        case QBP_DAMPED_SP:   return launch_one<1, MC, FUSED_MAX_THREADS>(P, cfg, s);
        default:              return launch_one<2, MC, FUSED_MAX_THREADS>(P, cfg, s);
    }
}

}  // namespace

hipError_t launch_fused(bool mc, int variant, const FusedParams& P, const LaunchCfg& cfg, hipStream_t s)
{
    return mc ? launch_variant<true>(variant, P, cfg, s) : launch_variant<false>(variant, P, cfg, s);
}

#ifndef QBP_FAST_TU
hipError_t launch_debug_math(int kind, const double* x, double* y, long long count, hipStream_t s)
{
    const int threads = 256;
    const long long blocks = (count + threads - 1) / threads;
    hipLaunchKernelGGL(debug_math_kernel, dim3((unsigned)blocks), dim3(threads), 0, s, kind, x, y, count);
    return hipGetLastError();
}

hipError_t launch_mc_sample(uint8_t* errors, int n, long long T, long long trial_begin, int draws,
                            unsigned long long seed, unsigned threshold, hipStream_t s)
{
    const long long items = T * ((n + 3) / 4);
    const int threads = 256;
    hipLaunchKernelGGL(mc_sample_kernel, dim3((unsigned)((items + threads - 1) / threads)), dim3(threads), 0, s,
                       errors, n, T, trial_begin, draws, seed, threshold);
    return hipGetLastError();
}
#endif  // QBP_FAST_TU

}  // namespace qbp
