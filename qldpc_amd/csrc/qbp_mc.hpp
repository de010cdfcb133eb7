// Monte-Carlo pieces shared by the kernels: the counter-based error sampler (this build's own
// specification, restated in oracle/bp_oracle.c:oracle_mc_errors) and the counter layout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qbp {

constexpr int NUM_COUNTERS = 12;

__device__ __forceinline__ void philox4x32_10(unsigned c[4], unsigned k0, unsigned k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const unsigned n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// Error bits of qubits 4g .. 4g+3 of trial `trial` as bytes (0/1) packed in a u32: one Philox
// evaluation per draw serves four qubits (specification: oracle/bp_oracle.c, oracle_mc_errors:
// counter = (trial lo, trial hi, qubit / 4, draw), word qubit % 4, bit = word < floor(p 2^32)).
__device__ __forceinline__ unsigned mc_error_quad(unsigned long long trial, int g, int draws,
                                                  unsigned long long seed, unsigned thr)
{
    unsigned bytes = 0;
    for (int d = 0; d < draws; ++d) {
        unsigned c[4] = {(unsigned)trial, (unsigned)(trial >> 32), (unsigned)g, (unsigned)d};
        philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
        bytes ^= (c[0] < thr ? 1u : 0u) | (c[1] < thr ? 0x100u : 0u) | (c[2] < thr ? 0x10000u : 0u) |
                 (c[3] < thr ? 0x1000000u : 0u);
    }
    return bytes;
}

// The same four bytes from a stored error pattern instead (errors [n] 0/1 bytes of one trial): the Monte-Carlo
// kernels run on given errors when FusedParams / GenericParams::errors_in is set (qbp_mc_run_errors: the
// reference-pinned classification test feeds the reference's own trials through the device pipeline).
__device__ __forceinline__ unsigned mc_stored_quad(const uint8_t* errors, int g, int n)
{
    unsigned bytes = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (4 * g + i < n) bytes |= (unsigned)(errors[4 * g + i] & 1u) << (8 * i);
    return bytes;
}

// Classification of one finished trial (paperResults_GPU.py:127-144 without the OSD call) into a
// counter row: lm = logical mask of hard ^ error, ew = weight of the error, df = hard != error.
__device__ __forceinline__ void mc_count_trial(int* cnt, unsigned long long lm, int ew, int df, int conv,
                                               int it, int half_distance)
{
    const bool logical = lm != 0ull;                       // (Lx @ residual) % 2 has a 1
    cnt[0] += 1;
    if (conv && !logical && df) cnt[5] += 1;               // degenerateErrors  (:134-135)
    if (logical) {
        cnt[1] += 1;                                       // logical_error     (:137-138)
        if (ew < half_distance) cnt[3] += 1; else cnt[4] += 1;   // (:140-144)
        if (!conv) cnt[8] += 1;
    }
    if (!conv) cnt[6] += 1;
    cnt[7] += it;
    if (!df) cnt[9] += 1;
}

}  // namespace qbp
