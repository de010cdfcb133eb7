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
// Rows and columns of H are processed by WEIGHT CLASS (rows of weight 1 .. 8, columns of weight
// 1 .. 4; the host sorts them, build_tables in qbp.hip): within a class the weight is a
// compile-time constant and the body is straight-line code -- no per-entry degree tests, which
// the compiler otherwise turns into branches around every load.  Neither step depends on the
// order in which rows (columns) are visited.  Longer rows / columns take plain loops.
//
// Arithmetic and order are those of the other kernels (ascending column within a row, ascending
// check within a column), so outputs are bit-identical to theirs (tested).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qbp_math.hpp"

// variables per group of the variable step (two groups are in flight)
#ifndef QBP_STREAM_VU
#define QBP_STREAM_VU 4
#endif

namespace qbp {

constexpr int STREAM_MAX_ROW_CLASS = 8;   // rows of weight 1 .. 8 have their own instantiation
constexpr int STREAM_MAX_COL_CLASS = 4;   // columns of weight 1 .. 4

struct StreamParams {
    int m, n, E;
    const uint8_t* syndromes;   // [B][m]
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
    // Checks sorted by row weight (stable): positions [row_off[k], row_off[k + 1]) of the sorted
    // check table hold the checks of weight k for k = 0 .. 8 and of weight > 8 for k = 9.
    int row_off[STREAM_MAX_ROW_CLASS + 3];
    // Variables sorted by column weight (stable): positions [col_off[k], col_off[k + 1]) hold the
    // variables of weight k for k = 0 .. 4 and of weight > 4 for k = 5; the edges of the weight-k
    // variables (k = 1 .. 4) are listed contiguously from col_edge_base[k] in the sorted edge table.
    int col_off[STREAM_MAX_COL_CLASS + 3];
    int col_edge_base[STREAM_MAX_COL_CLASS + 2];
};

// Check step for the checks of one row-weight class D (positions [begin, end) of the sorted check
// table; srow = check index, srow_e0 = its first edge).  beliefPropagation.py:114-126 /
// rework/decoding.py:28-56 with the row's values in registers.  The next row's Q values are
// requested before the current row's ~90 D FP64 instructions run, so their HBM latency is covered
// by arithmetic (software pipelining).
template <int VARIANT, int D>
__device__ __forceinline__ void stream_check_class(const double* Q, double* R, const uint8_t* synT,
                                                   long long ES, long long Bc,
                                                   const int32_t* __restrict__ srow,
                                                   const int32_t* __restrict__ srow_e0, int begin,
                                                   int end, double alpha, NpT np_tab)
{
    double qn[D];
    {
        const int e0 = srow_e0[begin];
#pragma unroll
        for (int j = 0; j < D; ++j) qn[j] = Q[(long long)(e0 + j) * ES];
    }
    for (int i = begin; i < end; ++i) {
        const int e0 = srow_e0[i];
        const unsigned sbit = synT[(long long)srow[i] * Bc];
        double q[D];
#pragma unroll
        for (int j = 0; j < D; ++j) q[j] = qn[j];
        if (i + 1 < end) {
            const int f0 = srow_e0[i + 1];
#pragma unroll
            for (int j = 0; j < D; ++j) qn[j] = Q[(long long)(f0 + j) * ES];
        }
        if constexpr (VARIANT == 2) {
            double sprod = 1.0, min1 = __builtin_inf(), min2 = __builtin_inf();
            int min1_j = -1;
            bool anynan = false;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                sprod *= q[j] < 0.0 ? -1.0 : 1.0;
                anynan |= q[j] != q[j];
                const double a = __builtin_fabs(q[j]);
                if (a < min1) { min1 = a; min1_j = j; }          // argmin: first occurrence
            }
            if (anynan) sprod = __builtin_nan("");               // np.sign(nan) = nan: whole row NaN
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double a = __builtin_fabs(q[j]);
                if (j != min1_j && a < min2) min2 = a;
            }
            const double as = sbit ? -alpha : alpha;             // alpha * syndrome_sign
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double sg = q[j] < 0.0 ? -1.0 : 1.0;
                const double mag = (__builtin_fabs(q[j]) == min1) ? min2 : min1;
                R[(long long)(e0 + j) * ES] = (as * (sprod * sg)) * mag;
            }
        } else {
            double t[D];
            double prod = 1.0;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                t[j] = tanh_half_msg<VARIANT>(q[j], np_tab);
                prod = (j == 0) ? t[0] : prod * t[j];            // np.prod, ascending column
            }
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double ts = __builtin_fabs(t[j]) < 1e-15 ? 1e-15 : t[j];
                const double r = check_message<VARIANT>(div_nr(prod, ts), sbit, np_tab);
                R[(long long)(e0 + j) * ES] = (VARIANT == 1) ? r * alpha : r;
            }
        }
    }
}

// Rows longer than STREAM_MAX_ROW_CLASS: plain two-pass form (R holds the tanh values in between).
template <int VARIANT>
__device__ __forceinline__ void stream_check_long(const double* Q, double* R, unsigned sbit,
                                                  long long ES, int e0, int deg, double alpha,
                                                  NpT np_tab)
{
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
        if (anynan) sprod = __builtin_nan("");
        for (int j = 0; j < deg; ++j) {
            const double a = __builtin_fabs(Q[(long long)(e0 + j) * ES]);
            if (j != min1_j && a < min2) min2 = a;
        }
        const double as = sbit ? -alpha : alpha;
        for (int j = 0; j < deg; ++j) {
            const double x = Q[(long long)(e0 + j) * ES];
            const double sg = x < 0.0 ? -1.0 : 1.0;
            const double mag = (__builtin_fabs(x) == min1) ? min2 : min1;
            R[(long long)(e0 + j) * ES] = (as * (sprod * sg)) * mag;
        }
    } else {
        double prod = 1.0;
        for (int j = 0; j < deg; ++j) {
            const double t = tanh_half_msg<VARIANT>(Q[(long long)(e0 + j) * ES], np_tab);
            R[(long long)(e0 + j) * ES] = t;
            prod = (j == 0) ? t : prod * t;
        }
        for (int j = 0; j < deg; ++j) {
            const double t = R[(long long)(e0 + j) * ES];
            const double ts = __builtin_fabs(t) < 1e-15 ? 1e-15 : t;
            const double r = check_message<VARIANT>(div_nr(prod, ts), sbit, np_tab);
            R[(long long)(e0 + j) * ES] = (VARIANT == 1) ? r * alpha : r;
        }
    }
}

// Variable step for the variables of one column-weight class D (positions [begin, end) of the
// sorted tables): value = sum of the D messages in ascending check order + prior, Q = value - R
// (beliefPropagation.py:129-133).  Pure streaming, 3 adds per message.  Two register sets: the
// gathers of group g + 1 are issued BEFORE the stores of group g -- vmcnt counts loads and stores
// in issue order, so waiting for gathers issued after a group's stores would also wait for those
// stores to be acknowledged.  The candidate error is only stored while the syndrome is undecided
// (once its outputs are frozen nothing reads it any more).
template <int VARIANT, int D>
__device__ __forceinline__ void stream_var_class(double* Q, const double* R, uint8_t* cand, long long ES,
                                                 long long Bc, const int32_t* __restrict__ svar,
                                                 const int32_t* __restrict__ sedge,
                                                 const double* __restrict__ prior, int begin, int end,
                                                 int edge_base, bool frozen, double damping,
                                                 double one_minus_damping, double clip_llr)
{
    constexpr int VU = QBP_STREAM_VU;
#define QBP_GATHER(r, i0)                                                                        \
    _Pragma("unroll") for (int u = 0; u < VU; ++u)                                               \
        _Pragma("unroll") for (int k = 0; k < D; ++k)                                            \
            r[u][k] = R[(long long)sedge[edge_base + ((i0) + u - begin) * D + k] * ES];
#define QBP_FINISH_ONE(r, i)                                                                     \
    {                                                                                            \
        const int v = svar[i];                                                                   \
        double sum = r[0];                                                                       \
        _Pragma("unroll") for (int k = 1; k < D; ++k) sum = sum + r[k];   /* ascending check */  \
        const double val = sum + prior[v];                                                       \
        _Pragma("unroll") for (int k = 0; k < D; ++k) {                                          \
            double* qp = Q + (long long)sedge[edge_base + ((i) - begin) * D + k] * ES;           \
            const double qnew = val - r[k];                                                      \
            if constexpr (VARIANT == 0) {                                                        \
                *qp = qnew;                                                                      \
            } else {                                                                             \
                const double x = damping * qnew + one_minus_damping * *qp;                       \
                const double y = x < -clip_llr ? -clip_llr : x;   /* np.clip, NaN stays */       \
                *qp = y > clip_llr ? clip_llr : y;                                               \
            }                                                                                    \
        }                                                                                        \
        if (!frozen) cand[(long long)v * Bc] = val < 0.0;                                        \
    }
#define QBP_FINISH(r, i0) _Pragma("unroll") for (int u = 0; u < VU; ++u) QBP_FINISH_ONE(r[u], (i0) + u)
    const int main_end = begin + (end - begin) / (2 * VU) * (2 * VU);
    if (main_end > begin) {
        double ra[VU][D], rb[VU][D];
        QBP_GATHER(ra, begin)
        for (int i = begin; i < main_end; i += 2 * VU) {
            QBP_GATHER(rb, i + VU)
            QBP_FINISH(ra, i)
            if (i + 2 * VU < main_end) { QBP_GATHER(ra, i + 2 * VU) }
            QBP_FINISH(rb, i + VU)
        }
    }
    for (int i = main_end; i < end; ++i) {
        double r1[D];
#pragma unroll
        for (int k = 0; k < D; ++k) r1[k] = R[(long long)sedge[edge_base + (i - begin) * D + k] * ES];
        QBP_FINISH_ONE(r1, i)
    }
#undef QBP_GATHER
#undef QBP_FINISH_ONE
#undef QBP_FINISH
}

// Parity of the candidate error against the syndrome for the checks of one row-weight class
// (beliefPropagation.py:137-139); returns 1 if some check of the class is unsatisfied.
template <int D>
__device__ __forceinline__ unsigned stream_parity_class(const uint8_t* cand, const uint8_t* synT,
                                                        long long Bc, const int32_t* __restrict__ srow,
                                                        const int32_t* __restrict__ srow_e0,
                                                        const int32_t* __restrict__ col_idx, int begin,
                                                        int end)
{
    unsigned unsat = 0;
    for (int i = begin; i < end; ++i) {
        const int e0 = srow_e0[i];
        unsigned par = synT[(long long)srow[i] * Bc];
        unsigned bits[D];
#pragma unroll
        for (int j = 0; j < D; ++j) bits[j] = cand[(long long)col_idx[e0 + j] * Bc];
#pragma unroll
        for (int j = 0; j < D; ++j) par ^= bits[j];
        unsat |= par;
    }
    return unsat;
}

// The index arrays of H and the priors are separate __restrict__ kernel arguments: only then can
// the compiler prove that the kernel's own stores do not alias them and fetch them with scalar
// loads (they are wave-uniform); through the struct -- or through the captures of a lambda -- they
// become per-lane vector loads with a full wait in front of every message access.
template <int VARIANT>
__global__ __launch_bounds__(256) void bp_stream_kernel(const StreamParams P,
                                                        const int32_t* __restrict__ g_col_idx,
                                                        const int32_t* __restrict__ g_col_ptr,
                                                        const int32_t* __restrict__ g_col_edge,
                                                        const double* __restrict__ g_prior,
                                                        const int32_t* __restrict__ g_srow,
                                                        const int32_t* __restrict__ g_srow_e0,
                                                        const int32_t* __restrict__ g_srow_deg,
                                                        const int32_t* __restrict__ g_svar,
                                                        const int32_t* __restrict__ g_sedge)
{
    // tables of tanh / arctanh (qbp_math.hpp): the kernel's only LDS use and its only barrier
    __shared__ __attribute__((aligned(16))) double np_lds[NP_LDS_DOUBLES];
    np_tables_to_lds(np_lds, threadIdx.x, blockDim.x);
    const NpT np_tab = lds_address(np_lds);
    __syncthreads();
    const long long lb = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long b = P.b0 + lb;
    const bool valid = lb < P.Bc && b < P.B;
    if (!valid) return;                      // whole trailing lanes only: no further barriers
    const int m = P.m, n = P.n;
    // Workspace layout [edge][syndrome]: neighbouring wavefronts touch neighbouring 512-byte lines,
    // which spreads every access wave over the HBM channels.  (A tile-major layout
    // [64 syndromes][edge][lane] was 4-10 % slower, with or without padding between the tiles.)
    const long long Bc = P.Bc;
    double* const Q = P.Q + lb;
    double* const R = P.R + lb;
    const long long ES = Bc;                 // stride between the rows of two edges
    uint8_t* const cand = P.cand + lb;
    uint8_t* const synT = P.synT + lb;
    const bool force_full = (P.flags & 1u) != 0;
    const double one_minus_damping = 1.0 - P.damping;
    constexpr int RC = STREAM_MAX_ROW_CLASS, CC = STREAM_MAX_COL_CLASS;

    for (int c = 0; c < m; ++c) synT[(long long)c * Bc] = P.syndromes[b * m + c] & 1u;
    for (int e = 0; e < P.E; ++e) Q[(long long)e * ES] = g_prior[g_col_idx[e]];   // Q = prior on edges

    bool frozen = false;
    int it = 0;
    for (; it < P.max_iter; ++it) {
        // ---- check step ---------------------------------------------------------------------
#define QBP_ROW_CLASS(D)                                                                         \
        if (P.row_off[D + 1] > P.row_off[D])                                                     \
            stream_check_class<VARIANT, D>(Q, R, synT, ES, Bc, g_srow, g_srow_e0, P.row_off[D],  \
                                           P.row_off[D + 1], P.alpha, np_tab);
        QBP_ROW_CLASS(1) QBP_ROW_CLASS(2) QBP_ROW_CLASS(3) QBP_ROW_CLASS(4)
        QBP_ROW_CLASS(5) QBP_ROW_CLASS(6) QBP_ROW_CLASS(7) QBP_ROW_CLASS(8)
#undef QBP_ROW_CLASS
        for (int i = P.row_off[RC + 1]; i < P.row_off[RC + 2]; ++i)
            stream_check_long<VARIANT>(Q, R, synT[(long long)g_srow[i] * Bc], ES, g_srow_e0[i],
                                       g_srow_deg[i], P.alpha, np_tab);

        // ---- variable step -------------------------------------------------------------------
        if (!frozen)
            for (int i = P.col_off[0]; i < P.col_off[1]; ++i) {        // variables without checks
                const int v = g_svar[i];
                cand[(long long)v * Bc] = g_prior[v] < 0.0;
            }
#define QBP_COL_CLASS(D)                                                                         \
        if (P.col_off[D + 1] > P.col_off[D])                                                     \
            stream_var_class<VARIANT, D>(Q, R, cand, ES, Bc, g_svar, g_sedge, g_prior, P.col_off[D], \
                                         P.col_off[D + 1], P.col_edge_base[D], frozen, P.damping, \
                                         one_minus_damping, P.clip_llr);
        QBP_COL_CLASS(1) QBP_COL_CLASS(2) QBP_COL_CLASS(3) QBP_COL_CLASS(4)
#undef QBP_COL_CLASS
        for (int i = P.col_off[CC + 1]; i < P.col_off[CC + 2]; ++i) {   // columns of weight > 4
            const int v = g_svar[i];
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

        // ---- syndrome check ------------------------------------------------------------------
        unsigned unsat = 0;
        if (!frozen) {
            for (int i = P.row_off[0]; i < P.row_off[1]; ++i)           // checks without variables
                unsat |= synT[(long long)g_srow[i] * Bc];
#define QBP_ROW_CLASS(D)                                                                         \
            if (P.row_off[D + 1] > P.row_off[D])                                                 \
                unsat |= stream_parity_class<D>(cand, synT, Bc, g_srow, g_srow_e0, g_col_idx,    \
                                                P.row_off[D], P.row_off[D + 1]);
            QBP_ROW_CLASS(1) QBP_ROW_CLASS(2) QBP_ROW_CLASS(3) QBP_ROW_CLASS(4)
            QBP_ROW_CLASS(5) QBP_ROW_CLASS(6) QBP_ROW_CLASS(7) QBP_ROW_CLASS(8)
#undef QBP_ROW_CLASS
            for (int i = P.row_off[RC + 1]; i < P.row_off[RC + 2]; ++i) {
                const int e0 = g_srow_e0[i], deg = g_srow_deg[i];
                unsigned par = synT[(long long)g_srow[i] * Bc];
                for (int j = 0; j < deg; ++j) par ^= cand[(long long)g_col_idx[e0 + j] * Bc];
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
