// Streaming belief propagation: one LANE per syndrome, messages in HBM.
//
// The on-chip kernel (qbp_kernels.hpp) keeps a syndrome's messages in registers/LDS and is bounded
// by the FP64 vector ALU; it needs m <= 1024 and narrow rows/columns.  This kernel is the other
// design point: for matrices of any size the messages live in
// global memory as structure-of-arrays [edge][syndrome], every lane of a wavefront decodes its own
// syndrome, and all indices of H are wave-uniform (scalar loads) -- so every message access is a
// fully coalesced 512-byte wave transaction and there are no barriers, no LDS, no atomics.  Per
// iteration and syndrome it moves exactly the algorithmic bytes of SURVEY 8(d): read Q, write R
// (check step), read R, write Q (variable step) = 4 * E * 8 bytes (5 * E * 8 for the damped
// variants, which also read the old Q) plus n + E bytes of hard-decision traffic: its HBM traffic
// measured by the PMC counters IS the roofline figure, not an effective bandwidth.
//
// Arithmetic and order are those of the other kernels (ascending column within a row, ascending
// check within a column), so outputs are bit-identical to theirs (tested).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qbp_math.hpp"

#ifndef QBP_STREAM_VU
#define QBP_STREAM_VU 8
#endif
// variables per group in the pipelined variable step of regular matrices
#ifndef QBP_STREAM_VU_REG
#define QBP_STREAM_VU_REG 4
#endif

namespace qbp {

struct StreamParams {
    int m, n, E;
    const int32_t* row_ptr;     // CSR
    const int32_t* col_idx;
    const int32_t* col_ptr;     // CSC: edge ids of each column, ascending check
    const int32_t* col_edge;
    const uint8_t* syndromes;   // [B][m]
    const double* prior;        // [n]
    long long B;                // syndromes in the whole call
    long long b0;               // first syndrome of this launch (chunk)
    long long Bc;               // lanes of this launch = row stride of the workspace arrays
    int max_iter;
    unsigned flags;
    double alpha, damping, clip_llr;
    uint8_t* hard;              // [B][n]
    uint8_t* converged;
    int32_t* iters;
    double* llr;                // [B][n]
    double* Q;                  // [E][Bc]
    double* R;                  // [E][Bc]
    uint8_t* cand;              // [n][Bc] candidate error of the current iteration
    uint8_t* synT;              // [m][Bc] syndromes, transposed once at the start
};

// DMAX: rows / columns of at most DMAX entries keep their working values in registers; longer ones
// fall back to an extra pass through memory (wave-uniform branch).
// The index arrays of H and the priors are separate __restrict__ kernel arguments: only then can
// the compiler prove that the kernel's own stores do not alias them and fetch them with scalar
// loads (they are wave-uniform); through the struct they became per-lane vector loads in front of
// every message access.
// DVR > 0: H is regular -- every row has exactly DMAX entries and every column exactly DVR (the BB
// codes of the reference: 6 and 3) -- and the body is straight-line code without degree tests.
template <int VARIANT, int DMAX, int DVR = 0>
__global__ __launch_bounds__(256) void bp_stream_kernel(const StreamParams P,
                                                        const int32_t* __restrict__ g_row_ptr,
                                                        const int32_t* __restrict__ g_col_idx,
                                                        const int32_t* __restrict__ g_col_ptr,
                                                        const int32_t* __restrict__ g_col_edge,
                                                        const double* __restrict__ g_prior)
{
    const long long lb = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long b = P.b0 + lb;
    const bool valid = lb < P.Bc && b < P.B;
    if (!valid) return;                      // whole trailing lanes only: no barriers in this kernel
    const int m = P.m, n = P.n;
    // Workspace layout [edge][syndrome]: neighbouring wavefronts touch neighbouring 512-byte lines,
    // which spreads every access wave over the HBM channels.  (A tile-major layout
    // [64 syndromes][edge][lane] was 10 % slower: all waves then walk their tiles at the same
    // offset and collide on the same channels.)
    constexpr bool REG = DVR > 0;
    const long long Bc = P.Bc;
    double* const Q = P.Q + lb;
    double* const R = P.R + lb;
    const long long ES = Bc;                 // stride between the rows of two edges
    uint8_t* const cand = P.cand + lb;
    uint8_t* const synT = P.synT + lb;
    const bool force_full = (P.flags & 1u) != 0;
    const double one_minus_damping = 1.0 - P.damping;

    for (int c = 0; c < m; ++c) synT[(long long)c * Bc] = P.syndromes[b * m + c] & 1u;
    for (int e = 0; e < P.E; ++e) Q[(long long)e * ES] = g_prior[g_col_idx[e]];   // Q = prior on edges

    bool frozen = false;
    int it = 0;
    for (; it < P.max_iter; ++it) {
        // ---- check step ---------------------------------------------------------------------
        // The next row's Q values are requested before the current row's ~500 FP64 instructions
        // run, so their HBM latency is covered by arithmetic (software pipelining; rows longer
        // than DMAX take the plain path below).
        double qn[DMAX];
        {
            const int e0 = REG ? 0 : g_row_ptr[0], deg = REG ? DMAX : g_row_ptr[1] - e0;
#pragma unroll
            for (int j = 0; j < DMAX; ++j) if (REG || j < deg) qn[j] = Q[(long long)(e0 + j) * ES];
        }
        for (int c = 0; c < m; ++c) {
            const int e0 = REG ? c * DMAX : g_row_ptr[c], deg = REG ? DMAX : g_row_ptr[c + 1] - e0;
            const unsigned sbit = synT[(long long)c * Bc];
            double q[DMAX];
#pragma unroll
            for (int j = 0; j < DMAX; ++j) q[j] = qn[j];
            if (c + 1 < m) {
                const int f0 = REG ? (c + 1) * DMAX : g_row_ptr[c + 1];
                const int fdeg = REG ? DMAX : g_row_ptr[c + 2] - f0;
#pragma unroll
                for (int j = 0; j < DMAX; ++j) if (REG || j < fdeg) qn[j] = Q[(long long)(f0 + j) * ES];
            }
            if (!REG && deg > DMAX) {
                // long row: plain two-pass form (R holds the tanh values in between)
                if constexpr (VARIANT == 2) {
                    double sprod = 1.0, min1 = __builtin_inf(), min2 = __builtin_inf();
                    int min1_j = -1;
                    bool anynan = false;
                    for (int j = 0; j < deg; ++j) {
                        const double x = Q[(long long)(e0 + j) * ES];
                        sprod *= x < 0.0 ? -1.0 : 1.0;
                        anynan |= x != x;
                        const double a = __builtin_fabs(x);
                        if (a < min1) { min1 = a; min1_j = j; }
                    }
                    if (anynan) sprod = __builtin_nan("");   // np.sign(nan) = nan: whole row NaN
                    for (int j = 0; j < deg; ++j) {
                        const double a = __builtin_fabs(Q[(long long)(e0 + j) * ES]);
                        if (j != min1_j && a < min2) min2 = a;
                    }
                    const double as = sbit ? -P.alpha : P.alpha;
                    for (int j = 0; j < deg; ++j) {
                        const double x = Q[(long long)(e0 + j) * ES];
                        const double sg = x < 0.0 ? -1.0 : 1.0;
                        const double mag = (__builtin_fabs(x) == min1) ? min2 : min1;
                        R[(long long)(e0 + j) * ES] = (as * (sprod * sg)) * mag;
                    }
                } else {
                    double prod = 1.0;
                    for (int j = 0; j < deg; ++j) {
                        const double t = tanh_half(Q[(long long)(e0 + j) * ES]);
                        R[(long long)(e0 + j) * ES] = t;
                        prod = (j == 0) ? t : prod * t;
                    }
                    for (int j = 0; j < deg; ++j) {
                        const double t = R[(long long)(e0 + j) * ES];
                        const double ts = __builtin_fabs(t) < 1e-15 ? 1e-15 : t;
                        double po = div_nr(prod, ts);
                        po = sbit ? -po : po;
                        const double r = atanh2(__builtin_fmin(__builtin_fmax(po, -0.9999999), 0.9999999));
                        R[(long long)(e0 + j) * ES] = (VARIANT == 1) ? r * P.alpha : r;
                    }
                }
                continue;
            }
            if constexpr (VARIANT == 2) {
                // rework/decoding.py:28-56
                double sprod = 1.0, min1 = __builtin_inf(), min2 = __builtin_inf();
                int min1_j = -1;
                bool anynan = false;
#pragma unroll
                for (int j = 0; j < DMAX; ++j) {
                    if (REG || j < deg) {
                        sprod *= q[j] < 0.0 ? -1.0 : 1.0;
                        anynan |= q[j] != q[j];
                        const double a = __builtin_fabs(q[j]);
                        if (a < min1) { min1 = a; min1_j = j; }
                    }
                }
                if (anynan) sprod = __builtin_nan("");       // np.sign(nan) = nan: whole row NaN
#pragma unroll
                for (int j = 0; j < DMAX; ++j) {
                    if (REG || j < deg) {
                        const double a = __builtin_fabs(q[j]);
                        if (j != min1_j && a < min2) min2 = a;
                    }
                }
                const double as = sbit ? -P.alpha : P.alpha;
#pragma unroll
                for (int j = 0; j < DMAX; ++j) {
                    if (REG || j < deg) {
                        const double sg = q[j] < 0.0 ? -1.0 : 1.0;
                        const double mag = (__builtin_fabs(q[j]) == min1) ? min2 : min1;
                        R[(long long)(e0 + j) * ES] = (as * (sprod * sg)) * mag;
                    }
                }
            } else {
                // beliefPropagation.py:114-126 with the row's tanh values in registers
                double t[DMAX];
                double prod = 1.0;
#pragma unroll
                for (int j = 0; j < DMAX; ++j) {
                    if (REG || j < deg) {
                        t[j] = tanh_half(q[j]);
                        prod = (j == 0) ? t[0] : prod * t[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < DMAX; ++j) {
                    if (REG || j < deg) {
                        const double ts = __builtin_fabs(t[j]) < 1e-15 ? 1e-15 : t[j];
                        double po = div_nr(prod, ts);
                        po = sbit ? -po : po;
                        const double r = atanh2(__builtin_fmin(__builtin_fmax(po, -0.9999999), 0.9999999));
                        R[(long long)(e0 + j) * ES] = (VARIANT == 1) ? r * P.alpha : r;
                    }
                }
            }
        }
        // ---- variable step -------------------------------------------------------------------
        // Pure streaming (3 adds per message): variables go in groups of VU with every gather of
        // the group issued before the first add, so VU * degree loads are in flight per lane.  The
        // candidate error is only stored while the syndrome is still undecided (once its outputs
        // are frozen nothing reads it any more).
        constexpr int VU = REG ? QBP_STREAM_VU_REG : QBP_STREAM_VU, DVF = REG ? DVR : 4;
        int v_begin = 0;
        if constexpr (REG) {
            // Two register sets: the gathers of group g + 1 are issued BEFORE the stores of group g.
            // (vmcnt counts loads and stores in issue order, so waiting for gathers that were
            // issued after a group's stores also waits for those stores to be acknowledged.)
#define QBP_GATHER(r, v0_)                                                                     \
            _Pragma("unroll") for (int u = 0; u < VU; ++u)                                     \
                _Pragma("unroll") for (int k = 0; k < DVF; ++k)                                \
                    r[u][k] = R[(long long)g_col_edge[((v0_) + u) * DVF + k] * ES];
#define QBP_FINISH(r, v0_)                                                                     \
            _Pragma("unroll") for (int u = 0; u < VU; ++u) {                                   \
                const int v = (v0_) + u;                                                       \
                double sum = r[u][0];                                                          \
                _Pragma("unroll") for (int k = 1; k < DVF; ++k) sum = sum + r[u][k];           \
                const double val = sum + g_prior[v];                                           \
                _Pragma("unroll") for (int k = 0; k < DVF; ++k) {                              \
                    double* qp = Q + (long long)g_col_edge[v * DVF + k] * ES;                  \
                    const double qnew = val - r[u][k];                                         \
                    if constexpr (VARIANT == 0) {                                              \
                        *qp = qnew;                                                            \
                    } else {                                                                   \
                        const double x = P.damping * qnew + one_minus_damping * *qp;           \
                        const double y = x < -P.clip_llr ? -P.clip_llr : x;                    \
                        *qp = y > P.clip_llr ? P.clip_llr : y;                                 \
                    }                                                                          \
                }                                                                              \
                if (!frozen) cand[(long long)v * Bc] = val < 0.0;                              \
            }
            const int n_main = n - n % (2 * VU);
            if (n_main > 0) {
                double ra[VU][DVF], rb[VU][DVF];
                QBP_GATHER(ra, 0)
                for (int v0 = 0; v0 < n_main; v0 += 2 * VU) {
                    QBP_GATHER(rb, v0 + VU)
                    QBP_FINISH(ra, v0)
                    if (v0 + 2 * VU < n_main) { QBP_GATHER(ra, v0 + 2 * VU) }
                    QBP_FINISH(rb, v0 + VU)
                }
            }
            v_begin = n_main;
#undef QBP_GATHER
#undef QBP_FINISH
        }
        for (int v0 = v_begin; v0 < n; v0 += VU) {
            bool narrow = true;
            if constexpr (!REG) {
#pragma unroll
                for (int u = 0; u < VU; ++u)
                    if (v0 + u < n) narrow = narrow && (g_col_ptr[v0 + u + 1] - g_col_ptr[v0 + u] <= DVF);
            }
            if (narrow) {
                double r[VU][DVF];
#pragma unroll
                for (int u = 0; u < VU; ++u) {
                    if (v0 + u < n) {
                        const int k0 = REG ? (v0 + u) * DVF : g_col_ptr[v0 + u];
                        const int deg = REG ? DVF : g_col_ptr[v0 + u + 1] - k0;
#pragma unroll
                        for (int k = 0; k < DVF; ++k)
                            if (REG || k < deg) r[u][k] = R[(long long)g_col_edge[k0 + k] * ES];
                    }
                }
#pragma unroll
                for (int u = 0; u < VU; ++u) {
                    if (v0 + u < n) {
                        const int v = v0 + u;
                        const int k0 = REG ? v * DVF : g_col_ptr[v];
                        const int deg = REG ? DVF : g_col_ptr[v + 1] - k0;
                        double sum = 0.0;
#pragma unroll
                        for (int k = 0; k < DVF; ++k)
                            if (REG || k < deg) sum = (k == 0) ? r[u][0] : sum + r[u][k];   // ascending check order
                        const double val = sum + g_prior[v];
#pragma unroll
                        for (int k = 0; k < DVF; ++k) {
                            if (REG || k < deg) {
                                double* qp = Q + (long long)g_col_edge[k0 + k] * ES;
                                const double qnew = val - r[u][k];
                                if constexpr (VARIANT == 0) {
                                    *qp = qnew;
                                } else {
                                    const double x = P.damping * qnew + one_minus_damping * *qp;
                                    const double y = x < -P.clip_llr ? -P.clip_llr : x;   // np.clip, NaN stays
                                    *qp = y > P.clip_llr ? P.clip_llr : y;
                                }
                            }
                        }
                        if (!frozen) cand[(long long)v * Bc] = val < 0.0;
                    }
                }
            } else {
                for (int v = v0; v < v0 + VU && v < n; ++v) {
                    const int k0 = g_col_ptr[v], deg = g_col_ptr[v + 1] - k0;
                    double sum = 0.0;
                    for (int k = 0; k < deg; ++k) {
                        const double rk = R[(long long)g_col_edge[k0 + k] * ES];
                        sum = (k == 0) ? rk : sum + rk;
                    }
                    const double val = sum + g_prior[v];
                    for (int k = 0; k < deg; ++k) {
                        const long long o = (long long)g_col_edge[k0 + k] * ES;
                        const double qnew = val - R[o];
                        if constexpr (VARIANT == 0) {
                            Q[o] = qnew;
                        } else {
                            const double x = P.damping * qnew + one_minus_damping * Q[o];
                            const double y = x < -P.clip_llr ? -P.clip_llr : x;
                            Q[o] = y > P.clip_llr ? P.clip_llr : y;
                        }
                    }
                    if (!frozen) cand[(long long)v * Bc] = val < 0.0;
                }
            }
        }
        // ---- syndrome check ------------------------------------------------------------------
        unsigned unsat = 0;
        if (!frozen) {
            for (int c = 0; c < m; ++c) {
                const int e0 = REG ? c * DMAX : g_row_ptr[c], deg = REG ? DMAX : g_row_ptr[c + 1] - e0;
                unsigned par = synT[(long long)c * Bc];
                if (REG || deg <= DMAX) {
                    unsigned bits[DMAX];
#pragma unroll
                    for (int j = 0; j < DMAX; ++j) if (REG || j < deg) bits[j] = cand[(long long)g_col_idx[e0 + j] * Bc];
#pragma unroll
                    for (int j = 0; j < DMAX; ++j) if (REG || j < deg) par ^= bits[j];
                } else {
                    for (int j = 0; j < deg; ++j) par ^= cand[(long long)g_col_idx[e0 + j] * Bc];
                }
                unsat |= par;
            }
        }
        const bool conv = !frozen && !unsat;
        const bool last = it == P.max_iter - 1;
        if (conv || (last && !frozen)) {
            // emit: values are recomputed from R (once per syndrome) instead of being stored n
            // doubles per iteration
            for (int v = 0; v < n; ++v) {
                const int k0 = g_col_ptr[v], k1 = g_col_ptr[v + 1];
                double s = 0.0;
                for (int k = k0; k < k1; ++k) {
                    const double rk = R[(long long)g_col_edge[k] * ES];
                    s = (k == k0) ? rk : s + rk;
                }
                const double val = s + g_prior[v];
                if (P.llr) P.llr[b * n + v] = val;
                if (P.hard) P.hard[b * n + v] = val < 0.0;
            }
            if (P.converged) P.converged[b] = conv;
            if (P.iters) P.iters[b] = it;
            frozen = true;
        }
        if (frozen && !force_full) break;     // lanes of a wave that are done idle until the last one is
    }
}

}  // namespace qbp
