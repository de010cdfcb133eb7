// Host-side launch interface between the API translation unit (qbp.hip) and the kernel translation
// units (qbp_tu_*.hip).  Each kernel family is compiled in its own translation unit so that the
// library builds in parallel (make -j) and a change to one kernel recompiles only its family; the
// API unit sees parameter structs and these prototypes, never a kernel instantiation.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace qbp {

struct FusedParams;
struct GenericParams;
struct StreamParams;
struct OsdParams;
struct OsdBigWorkspace;

// (row weight, column weight) shapes the on-chip kernel is instantiated for: every code of the
// reference's codes/ is (6, 3); (8, 4) covers their space-time matrices (spaceTime.py: row weight
// 6 + 2, column weight 3).  Anything wider, or with m > 1024, goes to the general-H kernel.
// Threads per workgroup of the on-chip kernel = its register budget: 1024 threads are 4 wavefronts per SIMD
// at <= 128 registers, 768 are 3 at <= 168 (build-time, A/B: tools/build_variants.sh).
#ifndef QBP_FUSED_MAX_THREADS
#define QBP_FUSED_MAX_THREADS 1024
#endif
constexpr int FUSED_MAX_THREADS = QBP_FUSED_MAX_THREADS;
constexpr int DC_SMALL = 6, DV_SMALL = 3;
constexpr int DC_WIDE = 8, DV_WIDE = 4;

struct LaunchCfg {
    int S, threads, lds_bytes, grid, slot_stride, dc;
    int one_barrier;        // forced-iteration decode launch with two copies of R in LDS (qbp_kernels.hpp)
    int r0_table;           // early-exit launch with the first check step's messages tabulated in LDS
};

// qbp_tu_fused.hip
hipError_t launch_fused(bool mc, int variant, const FusedParams& P, const LaunchCfg& cfg, hipStream_t s);
hipError_t launch_fused_fast_math(bool mc, int variant, const FusedParams& P, const LaunchCfg& cfg, hipStream_t s);
hipError_t launch_debug_math(int kind, const double* x, double* y, long long count, hipStream_t s);
hipError_t launch_mc_sample(uint8_t* errors, int n, long long T, long long trial_begin, int draws,
                            unsigned long long seed, unsigned threshold, hipStream_t s);
// qbp_tu_generic.hip
hipError_t launch_generic(bool mc, int mem, int variant, const GenericParams& G, int grid, int threads,
                          size_t lds, hipStream_t s);
hipError_t launch_permute_prior(const double* prior, const int32_t* svar, double* out, int n, hipStream_t s);
// qbp_tu_stream.hip
hipError_t launch_stream(int variant, unsigned grid, const StreamParams& P, const int32_t* col_idx,
                         const int32_t* col_ptr, const int32_t* col_edge, const double* prior,
                         const int32_t* srow, const int32_t* srow_e0, const int32_t* srow_deg,
                         const int32_t* svar, const int32_t* sedge, hipStream_t s);
// qbp_tu_osd.hip
hipError_t launch_osd_small(int words_per_row, unsigned grid, size_t lds, const OsdParams& O, hipStream_t s);
hipError_t launch_osd_big(unsigned grid, size_t lds, const OsdParams& O, const OsdBigWorkspace& Wk, hipStream_t s);
hipError_t launch_osd_blocked(int rows_per_thread, unsigned grid, size_t lds, const OsdParams& O,
                              const OsdBigWorkspace& Wk, hipStream_t s);
hipError_t launch_hist_minmax(int grid, const double* x, long long count, double* part, hipStream_t s);
hipError_t launch_hist_bin(int grid, size_t lds, const double* msg, const uint8_t* errors, const int32_t* col_idx,
                           long long B, int E, int n, const double* edges, int bins, unsigned long long* hist,
                           hipStream_t s);

}  // namespace qbp
