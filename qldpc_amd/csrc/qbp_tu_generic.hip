// libqbp.so, translation units of the general-H kernel (qbp_generic.hpp).  Compiled three times, once
// per memory mode (-DQBP_GENERIC_MEM=0 / 1 / 2: six instantiations each), so that the three builds run
// in parallel; the mode-0 unit also holds the dispatcher and the prior permutation kernel.
#ifndef QBP_GENERIC_MEM
#error "compile with -DQBP_GENERIC_MEM=0, 1 or 2"
#endif
#if QBP_GENERIC_MEM == 0
#define QBP_DEFINE_KERNELS 1
#endif
#include <hip/hip_runtime.h>

#include "../../include/qbp.h"
#include "qbp_generic.hpp"
#include "qbp_launch.hpp"

namespace qbp {
namespace {

template <int VARIANT, bool MC, int MEM>
hipError_t generic_launch_k(const GenericParams& G, int grid, int threads, size_t lds, hipStream_t s)
{
    auto kern = bp_generic_kernel<VARIANT, MC, MEM>;
    static thread_local size_t lds_set[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || lds_set[dev] < lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) lds_set[dev] = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, G);
    return hipGetLastError();
}

template <bool MC, int MEM>
hipError_t generic_launch_v(int variant, const GenericParams& G, int grid, int threads, size_t lds, hipStream_t s)
{
    switch (variant) {
        case QBP_SUM_PRODUCT: return generic_launch_k<0, MC, MEM>(G, grid, threads, lds, s);
        case QBP_DAMPED_SP:   return generic_launch_k<1, MC, MEM>(G, grid, threads, lds, s);
        default:              return generic_launch_k<2, MC, MEM>(G, grid, threads, lds, s);
    }
}

}  // namespace

#define QBP_CAT2(a, b) a##b
#define QBP_CAT(a, b) QBP_CAT2(a, b)

// launch_generic_mem0 / _mem1 / _mem2: this unit's memory mode
hipError_t QBP_CAT(launch_generic_mem, QBP_GENERIC_MEM)(bool mc, int variant, const GenericParams& G, int grid,
                                                        int threads, size_t lds, hipStream_t s)
{
    return mc ? generic_launch_v<true, QBP_GENERIC_MEM>(variant, G, grid, threads, lds, s)
              : generic_launch_v<false, QBP_GENERIC_MEM>(variant, G, grid, threads, lds, s);
}

#if QBP_GENERIC_MEM == 0
hipError_t launch_generic_mem1(bool, int, const GenericParams&, int, int, size_t, hipStream_t);
hipError_t launch_generic_mem2(bool, int, const GenericParams&, int, int, size_t, hipStream_t);

hipError_t launch_generic(bool mc, int mem, int variant, const GenericParams& G, int grid, int threads,
                          size_t lds, hipStream_t s)
{
    static_assert(GENERIC_MEM_GLOBAL == 0 && GENERIC_MEM_LDS == 1 && GENERIC_MEM_SPLIT == 2, "mode numbering");
    switch (mem) {
        case GENERIC_MEM_LDS:   return launch_generic_mem1(mc, variant, G, grid, threads, lds, s);
        case GENERIC_MEM_SPLIT: return launch_generic_mem2(mc, variant, G, grid, threads, lds, s);
        default:                return launch_generic_mem0(mc, variant, G, grid, threads, lds, s);
    }
}

hipError_t launch_permute_prior(const double* prior, const int32_t* svar, double* out, int n, hipStream_t s)
{
    hipLaunchKernelGGL(generic_permute_prior, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, prior, svar, out, n);
    return hipGetLastError();
}
#endif

}  // namespace qbp
