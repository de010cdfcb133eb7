// General-H belief propagation: any parity-check matrix (irregular degrees, wide rows, large m/n:
// space-time and circuit-level matrices).  One workgroup per
// syndrome (256 threads when there are enough syndromes to fill the chip, up to 1024 for the
// one-syndrome-per-call users of large matrices); messages live in a per-workgroup global-memory
// workspace (L2-resident for the sizes of interest) instead of registers/LDS.  Threads take checks
// in the check step (by weight class: coalesced, straight-line) and variables in the variable step;
// every product / sum runs sequentially in the reference's order (ascending column within a row: np.prod(axis=1); ascending check within a column: np.sum(axis=0)), so the
// arithmetic is the same as the fused kernel's and the oracle's.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qbp_math.hpp"
#include "qbp_mc.hpp"

namespace qbp {

constexpr int GENERIC_MAX_ROW_CLASS = 8;   // rows of weight 1 .. 8 have their own instantiation

struct GenericParams {
    int m, n, E;
    const int32_t* col_idx;     // CSR column indices
    const int32_t* col_ptr;     // CSC
    // Message layout of one syndrome ("class-blocked, transposed"): checks are sorted by row weight
    // (stable; build_tables in qbp.hip); the cnt checks of weight D occupy one block in which entry
    // j of the i-th such check sits at row_base[D] + j * cnt + i, so that consecutive threads (one
    // check each) touch consecutive doubles.  Checks of weight > 8 keep their entries contiguous,
    // behind the blocks: positions [row_base[9], E) = the "long" edges; their transcendental work is
    // done one thread per EDGE (long_edge_row = index of the check among the long ones), only the
    // sequential row product / minimum search runs one thread per check.
    const int32_t* srow;        // [m] check index, sorted by weight class
    const int32_t* srow_e0;     // [m] first CSR edge of that check
    const int32_t* srow_deg;    // [m] its weight
    const int32_t* epos;        // [E] CSR edge -> position in the layout
    const int32_t* cpos;        // [E] CSC slot (column-major, ascending check) -> position
    int row_off[GENERIC_MAX_ROW_CLASS + 3];    // class boundaries in srow (0 .. 8, > 8)
    int row_base[GENERIC_MAX_ROW_CLASS + 2];   // first position of each class block
    const int32_t* long_edge_row;   // [E - row_base[9]]
    double* wsL;                // [grid][3 * number of long checks] row product / (sprod, min1, min2)
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
    // Monte-Carlo mode (MC instantiation; paperResults_GPU.py:95-151 for any H): B trials starting
    // at global index trial_begin; errors from the counter-based sampler, syndrome = H e, decode,
    // classify against the logical operators, add to counters[NUM_COUNTERS]
    const unsigned long long* lx_cols;  // [n] bit l = Lx[l][v]
    long long trial_begin;
    unsigned long long seed;
    unsigned threshold;                 // floor(p * 2^32)
    int draws;
    int half_distance;
    long long* counters;
    uint8_t* wsE;                       // [grid][n4] sampled error of the current trial
    uint8_t* wsS;                       // [grid][m] its syndrome
};

// Check update of one row held in registers: q[D] -> r[D]   (beliefPropagation.py:114-126 /
// rework/decoding.py:28-56).  `scale` is false for the alpha_estimation dump of the damped variant
// (rework/decoding.py:168-169 returns R before the alpha scaling).
template <int VARIANT, int D>
__device__ __forceinline__ void generic_row_update(const double (&q)[D], double (&r)[D], unsigned sbit,
                                                   double alpha, bool scale)
{
    if constexpr (VARIANT == 2) {
        double sprod = 1.0, min1 = __builtin_inf(), min2 = __builtin_inf();
        int min1_j = -1;
        bool anynan = false;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            sprod *= q[j] < 0.0 ? -1.0 : 1.0;
            anynan |= q[j] != q[j];
            const double a = __builtin_fabs(q[j]);
            if (a < min1) { min1 = a; min1_j = j; }
        }
        // np.sign(nan) = nan: one NaN message makes the row's sign product, hence every R of the
        // row, NaN (rework/decoding.py:28-35; inf - inf with infinite priors)
        if (anynan) sprod = __builtin_nan("");
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double a = __builtin_fabs(q[j]);
            if (j != min1_j && a < min2) min2 = a;
        }
        const double as = sbit ? -alpha : alpha;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double sg = q[j] < 0.0 ? -1.0 : 1.0;
            const double mag = (__builtin_fabs(q[j]) == min1) ? min2 : min1;
            r[j] = (as * (sprod * sg)) * mag;
        }
    } else {
        double t[D];
        double prod = 1.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            t[j] = tanh_half(q[j]);
            prod = (j == 0) ? t[0] : prod * t[j];
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double ts = __builtin_fabs(t[j]) < 1e-15 ? 1e-15 : t[j];
            double po = div_nr(prod, ts);
            po = sbit ? -po : po;
            const double x = atanh2(__builtin_fmin(__builtin_fmax(po, -0.9999999), 0.9999999));
            r[j] = (VARIANT == 1 && scale) ? x * alpha : x;
        }
    }
}

// __launch_bounds__(1024) = at most 128 registers: also right for the 256-thread launches, which
// then fit 4 workgroups per CU (a 135-register build with 3 per CU was 30 % slower).
template <int VARIANT, bool MC = false>
__global__ __launch_bounds__(1024) void bp_generic_kernel(const GenericParams P)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int m = P.m, n = P.n, E = P.E;
    constexpr int RC = GENERIC_MAX_ROW_CLASS;
    double* Q = P.wsQ + (size_t)blockIdx.x * E;
    double* R = P.wsR + (size_t)blockIdx.x * E;
    double* V = P.wsV + (size_t)blockIdx.x * n;
    uint8_t* cand = P.wsC + (size_t)blockIdx.x * n;
    const bool force_full = (P.flags & 1u) != 0;
    const double one_minus_damping = 1.0 - P.damping;
    // Monte-Carlo mode: workgroup-wide accumulators of the trial being classified, and thread 0's
    // counter row for all trials of this workgroup
    __shared__ unsigned long long mc_lmask;
    __shared__ int mc_weight, mc_diff;
    int mc_cnt[NUM_COUNTERS] = {0};
    const int n4 = (n + 3) / 4;
    uint8_t* const err = MC ? P.wsE + (size_t)blockIdx.x * n4 * 4 : nullptr;
    if constexpr (MC) {
        if (tid == 0) { mc_lmask = 0ull; mc_weight = 0; mc_diff = 0; }
    }

    for (long long b = blockIdx.x; b < P.B; b += gridDim.x) {
        const uint8_t* syn = P.syndromes + b * m;
        if constexpr (MC) {
            // errors of trial trial_begin + b (one Philox evaluation per four qubits), then its
            // syndrome H e mod 2 (beliefPropagationGPU.py:195-198)
            uint8_t* const sy = P.wsS + (size_t)blockIdx.x * m;
            for (int g = tid; g < n4; g += nt)
                reinterpret_cast<unsigned*>(err)[g] =
                    mc_error_quad((unsigned long long)(P.trial_begin + b), g, P.draws, P.seed, P.threshold);
            __syncthreads();
            for (int i = tid; i < m; i += nt) {
                unsigned par = 0;
                const int e0 = P.srow_e0[i], deg = P.srow_deg[i];
                for (int j = 0; j < deg; ++j) par ^= err[P.col_idx[e0 + j]];
                sy[P.srow[i]] = (uint8_t)(par & 1u);
            }
            syn = sy;
            __syncthreads();
        }
        for (int e = tid; e < E; e += nt) Q[P.epos[e]] = P.prior[P.col_idx[e]];   // Q = prior on edges
        __syncthreads();
        bool frozen = false;
        int it = 0;
        // Monte-Carlo: classification of the candidate error in `cand` (paperResults_GPU.py:113-144),
        // called by the whole workgroup right after the barrier that made `cand` final
        auto classify = [&](int conv, int it_done) {
            unsigned long long lm = 0ull;
            int ew = 0, df = 0;
            for (int v = tid; v < n; v += nt) {
                const unsigned e = err[v];
                const unsigned res = (unsigned)cand[v] ^ e;
                ew += (int)e;
                df |= (int)res;
                if (res) lm ^= P.lx_cols[v];
            }
            if (lm) atomicXor(&mc_lmask, lm);
            if (ew) atomicAdd(&mc_weight, ew);
            if (df) atomicOr(&mc_diff, 1);
            __syncthreads();
            if (tid == 0) {
                mc_count_trial(mc_cnt, mc_lmask, mc_weight, mc_diff, conv, it_done, P.half_distance);
                mc_lmask = 0ull; mc_weight = 0; mc_diff = 0;
            }
        };
        (void)classify;
        for (; it < P.max_iter; ++it) {
            const bool scale = !(P.dump_R != nullptr && it == P.dump_iter);
            // ---- check step, one thread per check, by weight class: D coalesced loads, straight-
            //      line arithmetic, D coalesced stores
#define QBP_ROW_CLASS(D)                                                                          \
            {                                                                                     \
                const int cnt = P.row_off[D + 1] - P.row_off[D];                                  \
                for (int i = tid; i < cnt; i += nt) {                                             \
                    const unsigned sbit = syn[P.srow[P.row_off[D] + i]] & 1u;                      \
                    double q[D], r[D];                                                            \
                    _Pragma("unroll") for (int j = 0; j < D; ++j) q[j] = Q[P.row_base[D] + j * cnt + i]; \
                    generic_row_update<VARIANT, D>(q, r, sbit, P.alpha, scale);                   \
                    _Pragma("unroll") for (int j = 0; j < D; ++j) R[P.row_base[D] + j * cnt + i] = r[j]; \
                }                                                                                 \
            }
            QBP_ROW_CLASS(1) QBP_ROW_CLASS(2) QBP_ROW_CLASS(3) QBP_ROW_CLASS(4)
            QBP_ROW_CLASS(5) QBP_ROW_CLASS(6) QBP_ROW_CLASS(7) QBP_ROW_CLASS(8)
#undef QBP_ROW_CLASS
            // ---- checks of weight > 8: the per-edge work (tanh; division + atanh) one thread per
            //      edge, the sequential part (np.prod in ascending column order / argmin and second
            //      minimum) one thread per check, two workgroup barriers in between
            const int first_long = P.row_off[RC + 1], n_long = P.row_off[RC + 2] - first_long;
            if (n_long > 0) {                                           // uniform
                const int lbase = P.row_base[RC + 1], n_ledges = E - lbase;
                double* const L = P.wsL + (size_t)blockIdx.x * 3 * n_long;
                if constexpr (VARIANT != 2) {
                    for (int k = tid; k < n_ledges; k += nt) R[lbase + k] = tanh_half(Q[lbase + k]);
                    __syncthreads();
                }
                for (int i = tid; i < n_long; i += nt) {
                    const int deg = P.srow_deg[first_long + i];
                    const int p0 = P.epos[P.srow_e0[first_long + i]];   // entries contiguous from here
                    if constexpr (VARIANT == 2) {
                        double sprod = 1.0, min1 = __builtin_inf(), min2 = __builtin_inf();
                        int min1_j = -1;
                        bool anynan = false;
                        for (int j = 0; j < deg; ++j) {
                            const double x = Q[p0 + j];
                            sprod *= x < 0.0 ? -1.0 : 1.0;
                            anynan |= x != x;
                            const double a = __builtin_fabs(x);
                            if (a < min1) { min1 = a; min1_j = j; }
                        }
                        if (anynan) sprod = __builtin_nan("");
                        for (int j = 0; j < deg; ++j) {
                            const double a = __builtin_fabs(Q[p0 + j]);
                            if (j != min1_j && a < min2) min2 = a;
                        }
                        L[3 * i] = sprod; L[3 * i + 1] = min1; L[3 * i + 2] = min2;
                    } else {
                        // eight loads in flight, then the multiplications in ascending order
                        double prod = 1.0;
                        int j = 0;
                        for (; j + 8 <= deg; j += 8) {
                            double a[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) a[u] = R[p0 + j + u];
#pragma unroll
                            for (int u = 0; u < 8; ++u) prod = (j + u == 0) ? a[0] : prod * a[u];
                        }
                        for (; j < deg; ++j) prod = (j == 0) ? R[p0] : prod * R[p0 + j];
                        L[3 * i] = prod;
                    }
                }
                __syncthreads();
                for (int k = tid; k < n_ledges; k += nt) {
                    const int i = P.long_edge_row[k];
                    const unsigned sbit = syn[P.srow[first_long + i]] & 1u;
                    if constexpr (VARIANT == 2) {
                        const double x = Q[lbase + k];
                        const double sg = x < 0.0 ? -1.0 : 1.0;
                        const double mag = (__builtin_fabs(x) == L[3 * i + 1]) ? L[3 * i + 2] : L[3 * i + 1];
                        const double as = sbit ? -P.alpha : P.alpha;
                        R[lbase + k] = (as * (L[3 * i] * sg)) * mag;
                    } else {
                        const double t = R[lbase + k];
                        const double ts = __builtin_fabs(t) < 1e-15 ? 1e-15 : t;
                        double po = div_nr(L[3 * i], ts);
                        po = sbit ? -po : po;
                        const double x = atanh2(__builtin_fmin(__builtin_fmax(po, -0.9999999), 0.9999999));
                        R[lbase + k] = (VARIANT == 1 && scale) ? x * P.alpha : x;
                    }
                }
            }
            __syncthreads();
            if (!scale) {
                for (int e = tid; e < E; e += nt) P.dump_R[b * E + e] = R[P.epos[e]] / P.dump_div;
                frozen = true;           // nothing else is reported for this syndrome
                break;
            }
            // ---- variable step (:129-136), one thread per variable: value, candidate error and
            //      the new variable->check messages of its column
            for (int v = tid; v < n; v += nt) {
                double s = 0.0;
                const int k0 = P.col_ptr[v], k1 = P.col_ptr[v + 1];
                for (int k = k0; k < k1; ++k) {
                    const double r = R[P.cpos[k]];
                    s = (k == k0) ? r : s + r;                      // ascending check order
                }
                const double val = s + P.prior[v];
                V[v] = val;
                cand[v] = val < 0.0;
                for (int k = k0; k < k1; ++k) {
                    const int o = P.cpos[k];
                    const double qn = val - R[o];
                    if constexpr (VARIANT == 0) {
                        Q[o] = qn;
                    } else {
                        const double q = P.damping * qn + one_minus_damping * Q[o];
                        const double y = q < -P.clip_llr ? -P.clip_llr : q;     // np.clip, NaN stays NaN
                        Q[o] = y > P.clip_llr ? P.clip_llr : y;
                    }
                }
            }
            __syncthreads();
            // ---- syndrome check (:137-139) -----------------------------------------------------
            int unsat = 0;
            if (!frozen) {
                for (int i = tid; i < m; i += nt) {
                    unsigned par = syn[P.srow[i]] & 1u;
                    const int e0 = P.srow_e0[i], deg = P.srow_deg[i];
                    for (int j = 0; j < deg; ++j) par ^= cand[P.col_idx[e0 + j]];
                    unsat |= (int)par;
                }
            }
            const int any_unsat = __syncthreads_or(unsat);   // also orders Q for the next check step
            const bool conv = !frozen && !any_unsat;
            if (conv) {
                if constexpr (MC) {
                    classify(1, it);
                } else {
                    for (int v = tid; v < n; v += nt) {
                        if (P.llr) P.llr[b * n + v] = V[v];
                        if (P.hard) P.hard[b * n + v] = cand[v];
                    }
                    if (tid == 0) {
                        if (P.converged) P.converged[b] = 1;
                        if (P.iters) P.iters[b] = it;
                    }
                }
                frozen = true;
                if (!force_full) break;
            }
        }
        if (!frozen) {
            if constexpr (MC) {
                classify(0, P.max_iter - 1);
            } else {
                for (int v = tid; v < n; v += nt) {
                    if (P.llr) P.llr[b * n + v] = V[v];
                    if (P.hard) P.hard[b * n + v] = cand[v];
                }
                if (tid == 0) {
                    if (P.converged) P.converged[b] = 0;
                    if (P.iters) P.iters[b] = P.max_iter - 1;
                }
            }
        }
        __syncthreads();
    }
    if constexpr (MC) {
        if (tid == 0)
            for (int i = 0; i < NUM_COUNTERS; ++i)
                if (mc_cnt[i])
                    atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + i),
                              (unsigned long long)mc_cnt[i]);
    }
}

}  // namespace qbp
