// libqbp.so, translation unit of the OSD-0 kernels (qbp_osd.hpp) and the histogram kernels (qbp_hist.hpp).
#define QBP_DEFINE_KERNELS 1
#include <hip/hip_runtime.h>

#include "../../include/qbp.h"
#include "qbp_hist.hpp"
#include "qbp_launch.hpp"
#include "qbp_osd.hpp"

namespace qbp {

hipError_t launch_osd_small(int words_per_row, unsigned grid, size_t lds, const OsdParams& O, hipStream_t s)
{
    // row width (32-bit words incl. the syndrome word) as a template argument for the codes of the
    // reference: n = 72 / 90 / 108 -> 4 or 5, 144 -> 6, 288 -> 10
    switch (words_per_row) {
#define QBP_OSD_CASE(WW) case WW: hipLaunchKernelGGL(osd0_kernel<WW>, dim3(grid), dim3(64), lds, s, O); break
        QBP_OSD_CASE(2); QBP_OSD_CASE(3); QBP_OSD_CASE(4); QBP_OSD_CASE(5); QBP_OSD_CASE(6);
        QBP_OSD_CASE(7); QBP_OSD_CASE(8); QBP_OSD_CASE(9); QBP_OSD_CASE(10); QBP_OSD_CASE(11);
#undef QBP_OSD_CASE
        default: hipLaunchKernelGGL(osd0_kernel<0>, dim3(grid), dim3(64), lds, s, O);
    }
    return hipGetLastError();
}

hipError_t launch_osd_big(unsigned grid, size_t lds, const OsdParams& O, const OsdBigWorkspace& Wk, hipStream_t s)
{
    static thread_local size_t lds_set[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds > 0 && (dev < 0 || dev >= 64 || lds_set[dev] < lds)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(osd0_big_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) lds_set[dev] = lds;
    }
    hipLaunchKernelGGL(osd0_big_kernel, dim3(grid), dim3(256), lds, s, O, Wk);
    return hipGetLastError();
}

template <int RPT>
static hipError_t launch_osd_blocked_rpt(unsigned grid, size_t lds, const OsdParams& O, const OsdBigWorkspace& Wk,
                                         hipStream_t s)
{
    static thread_local size_t lds_set[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || lds_set[dev] < lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(osd0_blocked_kernel<RPT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) lds_set[dev] = lds;
    }
    hipLaunchKernelGGL(osd0_blocked_kernel<RPT>, dim3(grid), dim3(1024), lds, s, O, Wk);
    return hipGetLastError();
}

// rows per thread: m <= 1024 * rows_per_thread (1, 2, 4 or 8)
hipError_t launch_osd_blocked(int rows_per_thread, unsigned grid, size_t lds, const OsdParams& O,
                              const OsdBigWorkspace& Wk, hipStream_t s)
{
    switch (rows_per_thread) {
        case 1: return launch_osd_blocked_rpt<1>(grid, lds, O, Wk, s);
        case 2: return launch_osd_blocked_rpt<2>(grid, lds, O, Wk, s);
        case 4: return launch_osd_blocked_rpt<4>(grid, lds, O, Wk, s);
        case 8: return launch_osd_blocked_rpt<8>(grid, lds, O, Wk, s);
        default: return hipErrorInvalidValue;
    }
}

#ifdef QBP_OSD_TIMING
extern "C" int qbp_debug_osd_timing(unsigned long long* out, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_osd_timing), 8 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(g_osd_stat), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_osd_timing), z, sizeof(z));
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_osd_stat), z, sizeof(z));
    }
    return (int)e;
}
#endif

hipError_t launch_hist_minmax(int grid, const double* x, long long count, double* part, hipStream_t s)
{
    hipLaunchKernelGGL(hist_minmax_kernel, dim3(grid), dim3(256), 0, s, x, count, part);
    return hipGetLastError();
}

hipError_t launch_hist_bin(int grid, size_t lds, const double* msg, const uint8_t* errors, const int32_t* col_idx,
                           long long B, int E, int n, const double* edges, int bins, unsigned long long* hist,
                           hipStream_t s)
{
    hipLaunchKernelGGL(hist_bin_kernel, dim3(grid), dim3(256), lds, s, msg, errors, col_idx, B, E, n, edges, bins, hist);
    return hipGetLastError();
}

}  // namespace qbp
