// General-H belief propagation: any parity-check matrix (irregular degrees, wide rows, large m/n:
// space-time and circuit-level matrices).  One workgroup per
// syndrome (256 threads when there are enough syndromes to fill the chip, up to 1024 for the
// one-syndrome-per-call users of large matrices); messages live in a per-workgroup global-memory workspace (L2-resident for the sizes
// of interest) instead of registers/LDS.  Threads take checks in the check step and variables in
// the variable step; every product / sum runs sequentially in the reference's order (ascending
// column within a row: np.prod(axis=1); ascending check within a column: np.sum(axis=0)), so the
// arithmetic is the same as the fused kernel's and the oracle's.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qbp_math.hpp"

namespace qbp {

struct GenericParams {
    int m, n, E;
    const int32_t* row_ptr;     // CSR
    const int32_t* col_idx;
    const int32_t* col_ptr;     // CSC: edge ids of each column in ascending check order
    const int32_t* col_edge;
    const uint8_t* syndromes;
    const double* prior;
    long long B;
    int max_iter;
    unsigned flags;
    double alpha, damping, clip_llr;
    uint8_t* hard;
    uint8_t* converged;
    int32_t* iters;
    double* llr;
    // workspace, one slice per workgroup
    double* wsQ;                // [grid][E]
    double* wsR;                // [grid][E]
    double* wsV;                // [grid][n]
    uint8_t* wsC;               // [grid][n] candidate error
    // message dump (alpha_estimation=True of rework/decoding.py:58-59 and :168-169): after the check
    // step of iteration dump_iter, write the check->variable messages of every edge (CSR order)
    // to dump_R[b][E] and stop decoding that syndrome.  dump_scale divides (min-sum: 1/alpha).
    double* dump_R;
    int dump_iter;
    double dump_div;
};

template <int VARIANT>
__global__ __launch_bounds__(1024) void bp_generic_kernel(const GenericParams P)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int m = P.m, n = P.n, E = P.E;
    double* Q = P.wsQ + (size_t)blockIdx.x * E;
    double* R = P.wsR + (size_t)blockIdx.x * E;
    double* V = P.wsV + (size_t)blockIdx.x * n;
    uint8_t* cand = P.wsC + (size_t)blockIdx.x * n;
    const bool force_full = (P.flags & 1u) != 0;
    const double one_minus_damping = 1.0 - P.damping;

    for (long long b = blockIdx.x; b < P.B; b += gridDim.x) {
        const uint8_t* syn = P.syndromes + b * m;
        for (int e = tid; e < E; e += nt) Q[e] = P.prior[P.col_idx[e]];
        __syncthreads();
        bool frozen = false;
        int it = 0;
        for (; it < P.max_iter; ++it) {
            // ---- check step (beliefPropagation.py:114-126 / rework/decoding.py:28-56) ----------
            for (int c = tid; c < m; c += nt) {
                const int b0 = P.row_ptr[c], e1 = P.row_ptr[c + 1];
                const unsigned sbit = syn[c] & 1u;
                if constexpr (VARIANT == 2) {
                    double sprod = 1.0, min1 = __builtin_inf(), min2 = __builtin_inf();
                    int min1_e = -1;
                    bool anynan = false;
                    for (int e = b0; e < e1; ++e) {
                        const double q = Q[e];
                        sprod *= q < 0.0 ? -1.0 : 1.0;
                        anynan |= q != q;
                        const double a = __builtin_fabs(q);
                        if (a < min1) { min1 = a; min1_e = e; }
                    }
                    // np.sign(nan) = nan: one NaN message makes the row's sign product, hence every
                    // R of the row, NaN (rework/decoding.py:28-35; inf - inf with infinite priors)
                    if (anynan) sprod = __builtin_nan("");
                    for (int e = b0; e < e1; ++e) {
                        const double a = __builtin_fabs(Q[e]);
                        if (e != min1_e && a < min2) min2 = a;
                    }
                    const double as = sbit ? -P.alpha : P.alpha;
                    for (int e = b0; e < e1; ++e) {
                        const double q = Q[e];
                        const double s = q < 0.0 ? -1.0 : 1.0;
                        const double mag = (__builtin_fabs(q) == min1) ? min2 : min1;
                        R[e] = (as * (sprod * s)) * mag;
                    }
                } else {
                    double prod = 1.0;
                    for (int e = b0; e < e1; ++e) {
                        const double t = tanh_half(Q[e]);
                        R[e] = t;                                  // R holds tanh for now
                        prod = (e == b0) ? t : prod * t;
                    }
                    for (int e = b0; e < e1; ++e) {
                        const double t = R[e];
                        const double ts = __builtin_fabs(t) < 1e-15 ? 1e-15 : t;
                        double po = div_nr(prod, ts);
                        po = sbit ? -po : po;
                        const double r = atanh2(__builtin_fmin(__builtin_fmax(po, -0.9999999), 0.9999999));
                        // the alpha_estimation dump (rework/decoding.py:168-169) precedes the scaling
                        R[e] = (VARIANT == 1 && !(P.dump_R != nullptr && it == P.dump_iter)) ? r * P.alpha : r;
                    }
                }
            }
            __syncthreads();
            if (P.dump_R != nullptr && it == P.dump_iter) {
                for (int e = tid; e < E; e += nt) P.dump_R[b * E + e] = R[e] / P.dump_div;
                frozen = true;           // nothing else is reported for this syndrome
                break;
            }
            // ---- variable step (:129-136) ------------------------------------------------------
            for (int v = tid; v < n; v += nt) {
                double s = 0.0;
                const int k0 = P.col_ptr[v], k1 = P.col_ptr[v + 1];
                for (int k = k0; k < k1; ++k) {
                    const double r = R[P.col_edge[k]];
                    s = (k == k0) ? r : s + r;
                }
                const double val = s + P.prior[v];
                V[v] = val;
                cand[v] = val < 0.0;
            }
            __syncthreads();
            for (int e = tid; e < E; e += nt) {
                const double qn = V[P.col_idx[e]] - R[e];
                if constexpr (VARIANT == 0) {
                    Q[e] = qn;
                } else {
                    const double q = P.damping * qn + one_minus_damping * Q[e];
                    const double y = q < -P.clip_llr ? -P.clip_llr : q;     // np.clip, NaN stays NaN
                    Q[e] = y > P.clip_llr ? P.clip_llr : y;
                }
            }
            // ---- syndrome check (:137-139) -----------------------------------------------------
            int unsat = 0;
            if (!frozen) {
                for (int c = tid; c < m; c += nt) {
                    unsigned par = syn[c] & 1u;
                    for (int e = P.row_ptr[c]; e < P.row_ptr[c + 1]; ++e) par ^= cand[P.col_idx[e]];
                    unsat |= (int)par;
                }
            }
            const int any_unsat = __syncthreads_or(unsat);   // also orders Q for the next check step
            const bool conv = !frozen && !any_unsat;
            if (conv) {
                for (int v = tid; v < n; v += nt) {
                    if (P.llr) P.llr[b * n + v] = V[v];
                    if (P.hard) P.hard[b * n + v] = cand[v];
                }
                if (tid == 0) {
                    if (P.converged) P.converged[b] = 1;
                    if (P.iters) P.iters[b] = it;
                }
                frozen = true;
                if (!force_full) break;
            }
        }
        if (!frozen) {
            for (int v = tid; v < n; v += nt) {
                if (P.llr) P.llr[b * n + v] = V[v];
                if (P.hard) P.hard[b * n + v] = cand[v];
            }
            if (tid == 0) {
                if (P.converged) P.converged[b] = 0;
                if (P.iters) P.iters[b] = P.max_iter - 1;
            }
        }
        __syncthreads();
    }
}

}  // namespace qbp
