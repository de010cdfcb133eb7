// libqbp.so, translation unit of the streaming kernel (qbp_stream.hpp).
#define QBP_DEFINE_KERNELS 1
#include <hip/hip_runtime.h>

#include "../../include/qbp.h"
#include "qbp_launch.hpp"
#include "qbp_stream.hpp"

namespace qbp {

hipError_t launch_stream(int variant, unsigned grid, const StreamParams& P, const int32_t* col_idx,
                         const int32_t* col_ptr, const int32_t* col_edge, const double* prior,
                         const int32_t* srow, const int32_t* srow_e0, const int32_t* srow_deg,
                         const int32_t* svar, const int32_t* sedge, hipStream_t s)
{
#define QBP_STREAM_LAUNCH(V) hipLaunchKernelGGL((bp_stream_kernel<V>), dim3(grid), dim3(256), 0, s, P, \
        col_idx, col_ptr, col_edge, prior, srow, srow_e0, srow_deg, svar, sedge)
    switch (variant) {
        case QBP_SUM_PRODUCT: QBP_STREAM_LAUNCH(0); break;
        case QBP_DAMPED_SP:   QBP_STREAM_LAUNCH(1); break;
        default:              QBP_STREAM_LAUNCH(2); break;
    }
#undef QBP_STREAM_LAUNCH
    return hipGetLastError();
}

}  // namespace qbp
