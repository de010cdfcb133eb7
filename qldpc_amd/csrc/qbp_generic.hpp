// General-H belief propagation: any parity-check matrix (irregular degrees, wide rows, large m/n:
// space-time and circuit-level matrices).
//
// One workgroup per syndrome.  The 2E messages of the syndrome live
//   * in LDS when 16 E bytes (plus the small bookkeeping below) fit the CU's 160 KiB   (LDSMSG), or
//   * in a per-workgroup global workspace that stays in L2 / Infinity Cache.
// One BP iteration is TWO workgroup barriers:
//   check step     one thread per check; checks are sorted by weight and the work items of all
//                  weight classes form ONE index range in which every class is padded to whole
//                  wavefronts (a wavefront runs exactly one straight-line body, found by a scalar
//                  search): D coalesced loads, the check update, D coalesced stores ("class-blocked,
//                  transposed" layout, see GenericParams);
//   -- barrier A --
//   variable step  one thread per variable, variables sorted by column weight in the same way:
//                  gathers the column's messages (ascending check: np.sum(R, axis=0)), posterior
//                  value, new variable->check messages in place.  The syndrome test of the reference
//                  (H hard == s, beliefPropagation.py:137-139) is kept incrementally: every variable
//                  whose hard decision is 1 flips the parity bit of its checks in an LDS bit vector
//                  that started as the syndrome, and moves a counter of unsatisfied checks;
//   -- barrier B --
//   the counter is zero <=> converged.
// Sum-product without damping (VARIANT 0) keeps ONE array of E messages, updated in place: a slot holds the
// variable->check message before the check step and the check->variable message after it (a check reads its
// row's slots, then writes them; a variable reads its column's slots, then writes them).  Half the LDS -- or,
// for a matrix whose 2E messages did not fit (2592 x 7776: E doubles are 164.7 KB), nearly all traffic of the
// variable step in LDS instead of scattered 8-byte stores to L2.  The posterior values of the current
// iteration then go to a per-workgroup array V[n] (coalesced, sorted variable order) for the emission.
// The damped variants need last iteration's Q beside R: two arrays, as before.
// Posterior values and hard decisions are not stored per iteration: when a syndrome is emitted
// (first convergence, or the iteration limit) they are recomputed from R, which is still that
// iteration's.  Every product / sum runs sequentially in the reference's order (ascending column
// within a row: np.prod(axis=1); ascending check within a column), so the outputs are bit-identical
// to the on-chip kernel's and the streaming kernel's (tested).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qbp_math.hpp"
#include "qbp_mc.hpp"

namespace qbp {

constexpr int GENERIC_MAX_ROW_CLASS = 8;   // rows of weight 1 .. 8 have their own instantiation
constexpr int GENERIC_MAX_COL_CLASS = 4;   // columns of weight 1 .. 4 likewise
constexpr int GENERIC_PAIRWISE_LEVELS = 4; // numpy's pairwise sum recursion, unrolled this deep
constexpr int GENERIC_PAIRWISE_MAX_COL = 128 << GENERIC_PAIRWISE_LEVELS;

struct GenericParams {
    int m, n, E;
    // ---- checks, sorted by row weight (stable): sorted position w -> check srow[w] ------------
    // Message layout of one syndrome ("class-blocked, transposed"): the cnt checks of weight D
    // occupy one block in which entry j of the i-th such check sits at row_base[D] + j * cnt + i,
    // so that consecutive threads (one check each) touch consecutive doubles.  Checks of weight > 8
    // keep their entries contiguous behind the blocks: positions [row_base[9], E) = the "long"
    // edges; their transcendental work is done one thread per EDGE (long_edge_row = index of the
    // check among the long ones), only the sequential row product / minimum search per check.
    const int32_t* srow;        // [m] check index
    const int32_t* srow_e0;     // [m] first CSR edge of that check
    const int32_t* srow_deg;    // [m] its weight
    const int32_t* col_idx;     // CSR column indices (Monte-Carlo: syndrome of the sampled error)
    const int32_t* epos;        // [E] CSR edge -> position in the layout
    int row_off[GENERIC_MAX_ROW_CLASS + 3];    // class boundaries in srow (weights 0 .. 8, > 8)
    int row_base[GENERIC_MAX_ROW_CLASS + 2];   // first position of each class block
    // work items of the check step: the checks of weight 1 .. 8 in sorted order, each class padded to
    // whole wavefronts (rpad_off[D] = first item of class D, a multiple of 64; rpad_off[9] = total),
    // so that a wavefront never holds two classes (it would run both straight-line bodies)
    int rpad_off[GENERIC_MAX_ROW_CLASS + 2];
    const int32_t* long_edge_row;   // [E - row_base[9]]
    double* wsL;                // [grid][3 * number of long checks] row product / (sprod, min1, min2)
    unsigned long long* work_counter;   // zeroed before launch: index - grid of the next undecoded syndrome
    // ---- variables, sorted by column weight (stable): sorted position x -> variable svar[x] ----
    // vpos / vrow hold, per column entry, the message position and the SORTED position of the
    // entry's check, in the same blocked-transposed arrangement: entry j (ascending check) of the
    // i-th variable of weight D at col_base[D] + j * cnt + i; variables of weight > 4 keep their
    // entries contiguous from lcol_ptr[i] on.
    const int32_t* svar;        // [n]
    const int32_t* vpos;        // [E]
    const int32_t* vrow;        // [E]
    // the same two tables with every column's entries in another summation order, used at iteration 0 only
    // (QBP_FLAG_DENSE_F_COLSUM_ITER0: include/qbp.h); null = one order throughout
    const int32_t* vpos0;
    const int32_t* vrow0;
    const int32_t* lcol_ptr;    // [number of long columns + 1], absolute offsets into vpos / vrow
    int col_off[GENERIC_MAX_COL_CLASS + 3];    // class boundaries in svar (weights 0 .. 4, > 4)
    int col_base[GENERIC_MAX_COL_CLASS + 2];
    int cpad_off[GENERIC_MAX_COL_CLASS + 2];   // work items of the variable step, padded like rpad_off
    const double* prior_sorted; // [n] prior of the sorted variable x (permuted once per call)
    int lds_tables;             // 1: every workgroup keeps its own copy of vpos and prior_sorted in LDS
                                // (read once per kernel instead of once per iteration from L2)
    int r_split;                // GENERIC_MEM_SPLIT: doubles of R held in LDS (positions below it)
    // ---- the call ---------------------------------------------------------------------------------
    const uint8_t* syndromes;
    long long B;
    int max_iter;
    unsigned flags;             // QBP_FLAG_*: bit 0 force full, bit 2 numpy pairwise column sums
    double alpha, damping, clip_llr;
    uint8_t* hard;
    uint8_t* converged;
    int32_t* iters;
    double* llr;
    double* wsQ;                // [grid][E]   (messages in global memory only; not with the in-place update)
    double* wsR;                // [grid][E]
    double* wsV;                // [grid][n]   in-place update (VARIANT 0): posterior values of the current iteration
    // message dump (alpha_estimation=True of rework/decoding.py:58-59 and :168-169): after the check
    // step of iteration dump_iter, write the check->variable messages of every edge (CSR order)
    // to dump_R[b][E] and stop decoding that syndrome.  dump_div divides (min-sum: alpha).
    double* dump_R;
    int dump_iter;
    double dump_div;
    // ---- Monte-Carlo mode (MC instantiation; paperResults_GPU.py:95-151 for any H): B trials
    // starting at global index trial_begin; errors from the counter-based sampler, syndrome = H e,
    // decode, classify against the logical operators, add to counters[NUM_COUNTERS]
    const unsigned long long* lx_cols;  // [n] bit l = Lx[l][v]
    long long trial_begin;
    unsigned long long seed;
    unsigned threshold;                 // floor(p * 2^32)
    int draws;
    int half_distance;
    long long* counters;
    uint8_t* wsE;                       // [grid][n4] sampled error of the current trial
    const uint8_t* errors_in;           // optional [B][n]: these errors instead of sampled ones (qbp_mc_run_errors)
    // Monte-Carlo + OSD: records of the trials BP did not converge on, indexed by the trial's
    // position in this launch (same convention as the on-chip kernel, qbp_kernels.hpp)
    long long* fail_list;               // null = classify the BP output directly
    unsigned long long* fail_count;
    uint8_t* fail_syn;                  // [B][m]
    double* fail_llr;                   // [B][n]
    uint8_t* fail_hard;                 // [B][n]
    uint8_t* fail_err;                  // [B][n]
};

// Dynamic LDS of one workgroup: messages (LDSMSG) + syndrome bits + two parity buffers + counters
// (an even number of 32-bit words) + optional tables (prior_sorted [n] doubles, vpos [E] ints).
__host__ __device__ inline size_t generic_lds_words(int m)
{
    const size_t mw = ((size_t)m + 31) >> 5;
    return (((3 * mw + 2 + 1) & ~(size_t)1) + 4 + NUM_COUNTERS + 2 + 1) & ~(size_t)1;
}
__host__ __device__ inline size_t generic_lds_bytes(int m, int E, int n, bool lds_msgs, bool lds_tables,
                                                    int r_split = 0, bool inplace = false)
{
    return (size_t)NP_LDS_BYTES +      // tables of tanh / arctanh (qbp_math.hpp), at the start
           (lds_msgs ? (size_t)(inplace ? 8 : 16) * (size_t)E : (size_t)8 * (size_t)r_split) + generic_lds_words(m) * 4 +
           (lds_tables ? (size_t)8 * (size_t)n + (size_t)4 * (size_t)E : 0);
}

// Check update of one row held in registers: q[D] -> r[D]   (beliefPropagation.py:114-126 /
// rework/decoding.py:28-56).  `scale` is false for the alpha_estimation dump of the damped variant
// (rework/decoding.py:168-169 returns R before the alpha scaling).
// `put(j, value)` takes message j as soon as it exists (eight finished messages waiting for their stores are
// sixteen registers the wide rows do not have).
template <int VARIANT, int D, typename Put>
__device__ __forceinline__ void generic_row_update(const double (&q)[D], const Put& put, unsigned sbit,
                                                   double alpha, bool scale, NpT np_tab)
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
            put(j, (as * (sprod * sg)) * mag);
        }
    } else {
        double t[D];
        double prod = 1.0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            t[j] = tanh_half_msg<VARIANT>(q[j], np_tab);
            prod = (j == 0) ? t[0] : prod * t[j];
            QBP_EDGE_FENCE();
        }
        // (t_safe, :122: |t| <= 1, so a product of at least 1e-15 has no factor below it -- one wave-uniform
        // test instead of D compares and 2 D selects, as in the on-chip kernel)
        if (__builtin_amdgcn_ballot_w64(!(__builtin_fabs(prod) >= 1e-15)) != 0ull) {
#pragma unroll
            for (int j = 0; j < D; ++j) t[j] = __builtin_fabs(t[j]) < 1e-15 ? 1e-15 : t[j];
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double ts = t[j];
            const double x = check_message<VARIANT>(div_nr(prod, ts), sbit, np_tab);     // :123-126
            put(j, (VARIANT == 1 && scale) ? x * alpha : x);
            QBP_EDGE_FENCE();
        }
    }
}

// np.sum over the gathered column R[pos[0 .. n)] in numpy's pairwise order (the loop form of the
// reference, decoding/beliefPropagation.py:68; oracle/bp_oracle.c:np_pairwise_sum states the
// algorithm).  The recursion above 128 terms is unrolled LEVEL times (n <= 128 << LEVEL).
template <int LEVEL, typename RAcc>
__device__ __forceinline__ double np_pairwise_gather(const RAcc& R, const int32_t* pos, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = (i == 0) ? R(pos[0]) : res + R(pos[i]);
        return res;
    }
    if (LEVEL == 0 || n <= 128) {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = R(pos[j]);
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += R(pos[i + j]);
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += R(pos[i]);
        return res;
    }
    if constexpr (LEVEL > 0) {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_gather<LEVEL - 1, RAcc>(R, pos, n2) + np_pairwise_gather<LEVEL - 1, RAcc>(R, pos + n2, n - n2);
    }
    return 0.0;
}

// Column-weight class of the sorted variable x (weights 1 .. 4): weight D, number of variables in the
// class, offset of the variable's first entry in vpos / vrow.  Written with compile-time indices only:
// a dynamically indexed kernel-argument array would be copied to scratch memory.
__device__ __forceinline__ void generic_col_class(const GenericParams& P, int x, int& D, int& cnt, int& o)
{
    if (x < P.col_off[2])      { D = 1; cnt = P.col_off[2] - P.col_off[1]; o = P.col_base[1] + (x - P.col_off[1]); }
    else if (x < P.col_off[3]) { D = 2; cnt = P.col_off[3] - P.col_off[2]; o = P.col_base[2] + (x - P.col_off[2]); }
    else if (x < P.col_off[4]) { D = 3; cnt = P.col_off[4] - P.col_off[3]; o = P.col_base[3] + (x - P.col_off[3]); }
    else                       { D = 4; cnt = P.col_off[5] - P.col_off[4]; o = P.col_base[4] + (x - P.col_off[4]); }
}

#ifdef QBP_DEFINE_KERNELS   /* non-template kernels: defined in their translation unit only */
// Once per decode call: prior of the sorted variable x.
__global__ void generic_permute_prior(const double* prior, const int32_t* svar, double* out, int n)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n) out[x] = prior[svar[x]];
}

#endif  // QBP_DEFINE_KERNELS

// __launch_bounds__(1024) = at most 128 registers: also right for the smaller launches, which then
// fit several workgroups per CU.
// MEM: where one syndrome's 2E messages live --
//   GENERIC_MEM_GLOBAL  Q and R in the per-workgroup global workspace (L2 / Infinity Cache);
//   GENERIC_MEM_LDS     both in LDS (16 E bytes fit);
//   GENERIC_MEM_SPLIT   Q in the global workspace, the first r_split doubles of R in LDS and the rest in
//                       the workspace: R is written once and read once per iteration, so this removes up
//                       to half of the traffic of a kernel that is bound by it (2592 x 7776: E doubles
//                       are 896 bytes more than the LDS holds).  R is then reached through flat
//                       pointers picked per access.
constexpr int GENERIC_MEM_GLOBAL = 0, GENERIC_MEM_LDS = 1, GENERIC_MEM_SPLIT = 2;
#ifndef QBP_GENERIC_INPLACE      /* 0: two message arrays for every variant (A/B, tests of the host's sizing) */
#define QBP_GENERIC_INPLACE 1
#endif

#ifdef QBP_GEN_FAKE_Q       /* timing-only build: the check step's Q loads come from LDS (wrong results) */
#define QBP_GEN_QLOAD(pos) fake_q(pos)
#else
#define QBP_GEN_QLOAD(pos) Qload(pos)
#endif

template <int VARIANT, bool MC, int MEM>
__global__ __launch_bounds__(1024) void bp_generic_kernel(const GenericParams P)
{
    constexpr bool LDSMSG = MEM == GENERIC_MEM_LDS;
    extern __shared__ __attribute__((aligned(16))) double gsm_all[];
    constexpr NpT np_tab = 0u;          // = the LDS address of gsm_all (no static LDS in this kernel: checked below)
    double* const gsm = gsm_all + NP_LDS_DOUBLES;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (lds_address(gsm_all) != np_tab) __builtin_trap();
    np_tables_to_lds(gsm_all, tid, nt);           // (published by the first barrier of the syndrome loop)
    const int lane = tid & 63;
    const int m = P.m, n = P.n, E = P.E;
    constexpr int RC = GENERIC_MAX_ROW_CLASS, CC = GENERIC_MAX_COL_CLASS;
    constexpr bool INPLACE = VARIANT == 0 && QBP_GENERIC_INPLACE != 0;     // one message array (see the top)
    double* Q;
    double* R;
    unsigned* words;
    if constexpr (LDSMSG) {
        Q = gsm; R = INPLACE ? gsm : gsm + E; words = reinterpret_cast<unsigned*>(gsm + (INPLACE ? 1 : 2) * (size_t)E);
    } else {
        R = P.wsR + (size_t)blockIdx.x * E; Q = INPLACE ? R : P.wsQ + (size_t)blockIdx.x * E;
        words = reinterpret_cast<unsigned*>(gsm + (MEM == GENERIC_MEM_SPLIT ? P.r_split : 0));
    }
    double* const V = INPLACE ? P.wsV + (size_t)blockIdx.x * n : nullptr;
    // check->variable message at layout position `pos` (split mode: an LDS access or a global one,
    // under the lanes' own masks -- whole wavefronts on one side skip the other)
    auto Rload = [&](int pos) -> double {
        if constexpr (MEM == GENERIC_MEM_SPLIT) {
            // (the empty asm keeps the two sides apart: merged, they become ONE flat load through a
            // selected pointer -- slower, and this compiler's backend then fails on some variants of
            // the kernel with "Illegal instruction detected" around src_shared_base)
            double v;
            if (pos < P.r_split) { v = gsm[pos]; asm volatile(""); } else { v = R[pos]; asm volatile(""); }
            return v;
        } else {
            return R[pos];
        }
    };
#ifdef QBP_GEN_FAKE_Q
    auto fake_q = [&](int pos) -> double {
        if constexpr (MEM == GENERIC_MEM_SPLIT) return gsm[pos < P.r_split ? pos : pos - P.r_split];   // (LDS only)
        else return Q[pos];
    };
#endif
    auto Rstore = [&](int pos, double v) {
        if constexpr (MEM == GENERIC_MEM_SPLIT) {
            if (pos < P.r_split) { gsm[pos] = v; asm volatile(""); } else { R[pos] = v; asm volatile(""); }
        } else {
            R[pos] = v;
        }
    };
    // variable->check message at `pos`: the same slot as R's with the in-place update
    auto Qload = [&](int pos) -> double {
        if constexpr (INPLACE) return Rload(pos); else return Q[pos];
    };
    auto Qstore = [&](int pos, double v) {
        if constexpr (INPLACE) Rstore(pos, v); else Q[pos] = v;
    };
    const int mw = (m + 31) >> 5;
    unsigned* const synw = words;                   // [mw] syndrome bits, sorted check order
    unsigned* const par = words + mw;               // [2][mw] parity of H hard ^ s, by iteration parity
    int* const unsat = reinterpret_cast<int*>(words + 3 * mw);     // [2] number of set bits of par
    const int acc_off = (3 * mw + 2 + 1) & ~1;
    unsigned long long* const mc_lmask = reinterpret_cast<unsigned long long*>(words + acc_off);
    int* const mc_weight = reinterpret_cast<int*>(words + acc_off + 2);
    int* const mc_diff = mc_weight + 1;
    int* const mc_cnt = mc_diff + 1;                // [NUM_COUNTERS] this workgroup's counter row (MC)
    // index (minus gridDim.x) of the workgroup's next syndrome, drawn from the launch's work counter
    unsigned* const next_item = reinterpret_cast<unsigned*>(mc_cnt + NUM_COUNTERS);
    // tables of the variable step: from LDS when the launch reserved room for them, else from L2
    const double* prior_t = P.prior_sorted;
    const int32_t* vpos_t = P.vpos;
    // (never together with a split R -- the LDS is full then -- and kept out of that instantiation at
    // compile time: it then has no pointer that may or may not be an LDS address)
    if (MEM != GENERIC_MEM_SPLIT && P.lds_tables) {
        double* const pt = reinterpret_cast<double*>(words + generic_lds_words(m));
        int32_t* const vt = reinterpret_cast<int32_t*>(pt + n);
        for (int i = tid; i < n; i += nt) pt[i] = P.prior_sorted[i];
        for (int i = tid; i < E; i += nt) vt[i] = P.vpos[i];
        prior_t = pt; vpos_t = vt;
        __syncthreads();
    }

    const bool force_full = (P.flags & 1u) != 0;
    const bool pairwise = (P.flags & 4u) != 0;
    const double one_minus_damping = 1.0 - P.damping;
    const int n4 = (n + 3) / 4;
    uint8_t* const err = MC ? P.wsE + (size_t)blockIdx.x * n4 * 4 : nullptr;
    if constexpr (MC) {
        if (tid == 0) {
            *mc_lmask = 0ull; *mc_weight = 0; *mc_diff = 0;
            for (int i = 0; i < NUM_COUNTERS; ++i) mc_cnt[i] = 0;
        }
    }
    const int first_long = P.row_off[RC + 1], n_long = P.row_off[RC + 2] - first_long;
    const int lbase = P.row_base[RC + 1], n_ledges = E - lbase;
    const int first_lcol = P.col_off[CC + 1], n_lcol = P.col_off[CC + 2] - first_lcol;

    // new variable->check message of one edge (:133 / rework/decoding.py:65-66, :179-181)
    auto q_update = [&](int o, double val, double r) {
        const double qn = val - r;
        if constexpr (VARIANT == 0) {
            Qstore(o, qn);
        } else {
            const double q = P.damping * qn + one_minus_damping * Q[o];
            const double y = q < -P.clip_llr ? -P.clip_llr : q;     // np.clip, NaN stays NaN
            Q[o] = y > P.clip_llr ? P.clip_llr : y;
        }
    };

    // Work distribution: the first syndrome of a workgroup is static, the following ones come from a
    // global counter (syndromes differ widely in iterations with early exit: a static stride leaves
    // the workgroups that drew the slow ones running alone at the end).  Thread 0 issues the atomic
    // when a syndrome starts and hands the result over when it ends, so its latency is never waited for.
    // (32 bits: one register instead of two across the whole decode; the host keeps a launch below
    // 2^31.  The split-R build has no register to spare and milliseconds per syndrome: it fetches at
    // the end and waits the microsecond.)
    constexpr bool EARLY_FETCH = MEM != GENERIC_MEM_SPLIT;
    const bool dynamic = P.B > (long long)gridDim.x;      // else every syndrome is a workgroup's first
    unsigned fetched = 0x7fffffffu;
    for (long long b = blockIdx.x; b < P.B;) {
        const uint8_t* const syn = MC ? nullptr : P.syndromes + b * m;
        if (tid == 0) {
            unsat[0] = 0;
            if constexpr (EARLY_FETCH)
                if (dynamic) fetched = atomicAdd(reinterpret_cast<unsigned*>(P.work_counter), 1u);
        }
        if constexpr (MC) {
            // errors of trial trial_begin + b: one Philox evaluation per four qubits
            // (beliefPropagationGPU.py:195)
            for (int g = tid; g < n4; g += nt)
                reinterpret_cast<unsigned*>(err)[g] =
                    P.errors_in ? mc_stored_quad(P.errors_in + b * n, g, n)
                                : mc_error_quad((unsigned long long)(P.trial_begin + b), g, P.draws, P.seed, P.threshold);
        }
        __syncthreads();      // (also: the previous syndrome's last readers of LDS are done)
        // ---- syndrome bits in sorted check order; parity buffer 0 := syndrome -------------------
        {
            int cnt = 0;
            for (int w0 = tid - lane; w0 < m; w0 += nt) {           // w0 is wave-uniform
                const int w = w0 + lane;
                unsigned bit = 0;
                if (w < m) {
                    if constexpr (MC) {                             // syndrome = H e mod 2 (:198)
                        const int e0 = P.srow_e0[w], deg = P.srow_deg[w];
                        for (int j = 0; j < deg; ++j) bit ^= err[P.col_idx[e0 + j]];
                        bit &= 1u;
                    } else {
                        bit = syn[P.srow[w]] & 1u;
                    }
                }
                const unsigned long long mask = __ballot(bit != 0);
                if (lane == 0) {
                    const int wi = w0 >> 5;
                    synw[wi] = (unsigned)mask; par[wi] = (unsigned)mask;
                    if (wi + 1 < mw) { synw[wi + 1] = (unsigned)(mask >> 32); par[wi + 1] = (unsigned)(mask >> 32); }
                    cnt += __builtin_popcountll(mask);
                }
            }
            if (lane == 0 && cnt) atomicAdd(&unsat[0], cnt);
        }
        // ---- Q = prior on the edges (beliefPropagation.py:107) -------------------------------------
        for (int x = tid + P.col_off[1]; x < first_lcol; x += nt) {
            int D, cnt, o;
            generic_col_class(P, x, D, cnt, o);
            const int32_t* const pos = vpos_t + o;
            const double pv = prior_t[x];
            for (int j = 0; j < D; ++j) Qstore(pos[(size_t)j * cnt], pv);
        }
        for (int i = tid; i < n_lcol; i += nt) {
            const double pv = prior_t[first_lcol + i];
            for (int k = P.lcol_ptr[i]; k < P.lcol_ptr[i + 1]; ++k) Qstore(vpos_t[k], pv);
        }
        __syncthreads();
        const int syn_weight = unsat[0];      // unsatisfied checks of the all-zero candidate

        // column tables of the current iteration (another order at iteration 0: GenericParams::vpos0)
        const int32_t* vp = vpos_t;
        const int32_t* vr = P.vrow;
        // Posterior value of a long column (weight > 4) from the current R
        auto long_column_value = [&](int i, const int32_t*& pos, int& deg) {
            const int k0 = P.lcol_ptr[i];
            deg = P.lcol_ptr[i + 1] - k0;
            pos = vp + k0;
            double s = 0.0;
            if (pairwise && deg >= 8) {
                s = np_pairwise_gather<GENERIC_PAIRWISE_LEVELS>(Rload, pos, deg);
            } else {
                for (int j = 0; j < deg; ++j) {
                    const double r = Rload(pos[j]);
                    s = (j == 0) ? r : s + r;                     // ascending check order
                }
            }
            return s + prior_t[first_lcol + i];
        };

        // Outputs of this syndrome from the current R (values, hard decisions): decode mode writes
        // them, Monte-Carlo mode classifies them (paperResults_GPU.py:113-144) or leaves the trial
        // to OSD-0.  Called by the whole workgroup.
        auto emit = [&](int conv, int it_done) {
            const bool to_osd = MC && P.fail_list != nullptr && !conv;
            unsigned long long lm = 0ull;
            int ew = 0, df = 0;
            for (int x = tid; x < n; x += nt) {
                double val;
                if (x < P.col_off[1]) {
                    val = prior_t[x];                        // isolated variable
                } else if (INPLACE) {
                    val = V[x];                              // (the slots hold Q by now)
                } else if (x < first_lcol) {
                    int D, cnt, o;
                    generic_col_class(P, x, D, cnt, o);
                    const int32_t* const pos = vp + o;
                    double s = 0.0;
                    for (int j = 0; j < D; ++j) {
                        const double r = Rload(pos[(size_t)j * cnt]);
                        s = (j == 0) ? r : s + r;
                    }
                    val = s + prior_t[x];
                } else {
                    const int32_t* pos; int deg;
                    val = long_column_value(x - first_lcol, pos, deg);
                }
                const int v = P.svar[x];
                const unsigned hd = val < 0.0 ? 1u : 0u;
                if constexpr (MC) {
                    const unsigned e = err[v];
                    if (to_osd) {
                        P.fail_llr[b * n + v] = val;
                        P.fail_hard[b * n + v] = (uint8_t)hd;
                        P.fail_err[b * n + v] = (uint8_t)e;
                    } else {
                        const unsigned res = hd ^ e;
                        ew += (int)e;
                        df |= (int)res;
                        if (res) lm ^= P.lx_cols[v];
                    }
                } else {
                    if (P.llr) P.llr[b * n + v] = val;
                    if (P.hard) P.hard[b * n + v] = (uint8_t)hd;
                }
            }
            if constexpr (MC) {
                if (to_osd) {
                    for (int w = tid; w < m; w += nt)
                        P.fail_syn[b * m + P.srow[w]] = (uint8_t)((synw[w >> 5] >> (w & 31)) & 1u);
                    if (tid == 0) {
                        P.fail_list[atomicAdd(P.fail_count, 1ull)] = b;
                        mc_cnt[0] += 1; mc_cnt[6] += 1; mc_cnt[7] += it_done;   // BP bookkeeping only
                    }
                } else {
                    if (lm) atomicXor(mc_lmask, lm);
                    if (ew) atomicAdd(mc_weight, ew);
                    if (df) atomicOr(mc_diff, 1);
                    __syncthreads();
                    if (tid == 0) {
                        mc_count_trial(mc_cnt, *mc_lmask, *mc_weight, *mc_diff, conv, it_done, P.half_distance);
                        *mc_lmask = 0ull; *mc_weight = 0; *mc_diff = 0;
                    }
                }
            } else if (tid == 0) {
                if (P.converged) P.converged[b] = (uint8_t)conv;
                if (P.iters) P.iters[b] = it_done;
            }
        };

        bool frozen = false;
        for (int it = 0; it < P.max_iter; ++it) {
            const bool scale = !(P.dump_R != nullptr && it == P.dump_iter);
            // ================= check step =======================================================
#ifndef QBP_GEN_SKIP_CHECK      /* (timing-only builds: tools/build_variants.sh) */
            for (int wp0 = tid - lane; wp0 < P.rpad_off[RC + 1]; wp0 += nt) {
                // one wavefront = 64 consecutive work items of ONE weight class (scalar class search)
                const int wpu = __builtin_amdgcn_readfirstlane(wp0);
                int D = 1;              // (compile-time indices only: a dynamically indexed kernel
#pragma unroll                          //  argument array would be copied to scratch memory)
                for (int k = 2; k <= RC; ++k) D += wpu >= P.rpad_off[k] ? 1 : 0;
                // (opaque: keeps the per-class address arithmetic inside the loop.  Hoisted, it is ~60
                // loop-invariant registers that get spilled to scratch.)
                int lane_ = lane;
                asm volatile("" : "+v"(lane_));
#define QBP_ROW_CLASS(DD)                                                                          \
                case DD: {                                                                         \
                    const int cnt = P.row_off[DD + 1] - P.row_off[DD];                             \
                    const int i = wpu - P.rpad_off[DD] + lane_;                                    \
                    if (i < cnt) {                                                                 \
                        const int w = P.row_off[DD] + i;                                           \
                        const unsigned sbit = (synw[w >> 5] >> (w & 31)) & 1u;                     \
                        const int base = P.row_base[DD] + i;                                       \
                        double q[DD];                                                              \
                        _Pragma("unroll") for (int j = 0; j < DD; ++j) q[j] = QBP_GEN_QLOAD(base + j * cnt);   \
                        generic_row_update<VARIANT, DD>(q, [&](int j, double v) { Rstore(base + j * cnt, v); }, \
                                                        sbit, P.alpha, scale, np_tab);                     \
                    }                                                                              \
                } break;
                switch (D) {
                    QBP_ROW_CLASS(1) QBP_ROW_CLASS(2) QBP_ROW_CLASS(3) QBP_ROW_CLASS(4)
                    QBP_ROW_CLASS(5) QBP_ROW_CLASS(6) QBP_ROW_CLASS(7) QBP_ROW_CLASS(8)
                    default: break;
                }
#undef QBP_ROW_CLASS
            }
#endif
            // ---- checks of weight > 8: the per-edge work (tanh; division + atanh) one thread per
            //      edge, the sequential part (np.prod in ascending column order / argmin and second
            //      minimum) one thread per check, two more workgroup barriers in between
            if (n_long > 0) {                                           // uniform
                double* const L = P.wsL + (size_t)blockIdx.x * 3 * n_long;
                if constexpr (VARIANT != 2) {
                    for (int k = tid; k < n_ledges; k += nt) Rstore(lbase + k, tanh_half_msg<VARIANT>(Qload(lbase + k), np_tab));
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
                            for (int u = 0; u < 8; ++u) a[u] = Rload(p0 + j + u);
#pragma unroll
                            for (int u = 0; u < 8; ++u) prod = (j + u == 0) ? a[0] : prod * a[u];
                        }
                        for (; j < deg; ++j) prod = (j == 0) ? Rload(p0) : prod * Rload(p0 + j);
                        L[3 * i] = prod;
                    }
                }
                __syncthreads();
                for (int k = tid; k < n_ledges; k += nt) {
                    const int i = P.long_edge_row[k];
                    const int w = first_long + i;
                    const unsigned sbit = (synw[w >> 5] >> (w & 31)) & 1u;
                    if constexpr (VARIANT == 2) {
                        const double x = Q[lbase + k];
                        const double sg = x < 0.0 ? -1.0 : 1.0;
                        const double mag = (__builtin_fabs(x) == L[3 * i + 1]) ? L[3 * i + 2] : L[3 * i + 1];
                        const double as = sbit ? -P.alpha : P.alpha;
                        Rstore(lbase + k, (as * (L[3 * i] * sg)) * mag);
                    } else {
                        const double t = Rload(lbase + k);
                        const double ts = __builtin_fabs(t) < 1e-15 ? 1e-15 : t;
                        const double x = check_message<VARIANT>(div_nr(L[3 * i], ts), sbit, np_tab);
                        Rstore(lbase + k, (VARIANT == 1 && scale) ? x * P.alpha : x);
                    }
                }
            }
            __syncthreads();                                          // ---- barrier A
            if (!scale) {
                for (int e = tid; e < E; e += nt) P.dump_R[b * E + e] = Rload(P.epos[e]) / P.dump_div;
                frozen = true;           // nothing else is reported for this syndrome
                break;
            }
            // ================= variable step (:129-136) + incremental syndrome test (:137-139) ====
            vp = (it == 0 && P.vpos0) ? P.vpos0 : vpos_t;
            vr = (it == 0 && P.vrow0) ? P.vrow0 : P.vrow;
            const int p = it & 1;
            unsigned* const pbuf = par + p * mw;
            {   // the other buffer becomes the syndrome again (its last readers passed barrier A)
                unsigned* const obuf = par + (p ^ 1) * mw;
                for (int i = tid; i < mw; i += nt) obuf[i] = synw[i];
                if (tid == 0) unsat[p ^ 1] = syn_weight;
            }
            int delta = 0;
            auto flip = [&](int cw) {             // the check at sorted position cw changes parity
                const unsigned bit = 1u << (cw & 31);
                const unsigned old = atomicXor(&pbuf[cw >> 5], bit);
                delta += (old & bit) ? -1 : 1;
            };
#ifndef QBP_GEN_SKIP_VAR
            // (Measured and dropped: a software-pipelined form that requests the next pass's positions
            // and prior before working on the current pass -- no gain on any matrix.  Where the messages
            // live in L2 the step is bound by the rate of its scattered 8-byte stores of Q, one cache
            // line per cycle and CU: 20 592 of them are the 8 us the step takes on 2592 x 7776.)
            for (int xp0 = tid - lane; xp0 < P.cpad_off[CC + 1]; xp0 += nt) {
                const int xpu = __builtin_amdgcn_readfirstlane(xp0);
                int D = 1;
#pragma unroll
                for (int k = 2; k <= CC; ++k) D += xpu >= P.cpad_off[k] ? 1 : 0;
                int lane_ = lane;
                asm volatile("" : "+v"(lane_));
#define QBP_COL_CLASS(DD)                                                                          \
                case DD: {                                                                         \
                    const int cnt = P.col_off[DD + 1] - P.col_off[DD];                             \
                    const int i = xpu - P.cpad_off[DD] + lane_;                                    \
                    if (i < cnt) {                                                                 \
                        const int base = P.col_base[DD] + i;                                       \
                        int o[DD];                                                                 \
                        double r[DD];                                                              \
                        _Pragma("unroll") for (int j = 0; j < DD; ++j) o[j] = vp[base + j * cnt];    \
                        _Pragma("unroll") for (int j = 0; j < DD; ++j) r[j] = Rload(o[j]);         \
                        double s = r[0];                                                           \
                        _Pragma("unroll") for (int j = 1; j < DD; ++j) s = s + r[j];               \
                        const double val = s + prior_t[P.col_off[DD] + i];                         \
                        if constexpr (INPLACE) V[P.col_off[DD] + i] = val;                         \
                        if (!frozen && val < 0.0) {                                                \
                            _Pragma("unroll") for (int j = 0; j < DD; ++j) flip(vr[base + j * cnt]); \
                        }                                                                          \
                        _Pragma("unroll") for (int j = 0; j < DD; ++j) q_update(o[j], val, r[j]);  \
                    }                                                                              \
                } break;
                switch (D) {
                    QBP_COL_CLASS(1) QBP_COL_CLASS(2) QBP_COL_CLASS(3) QBP_COL_CLASS(4)
                    default: break;
                }
#undef QBP_COL_CLASS
            }
#endif
            for (int i = tid; i < n_lcol; i += nt) {
                const int32_t* pos; int deg;
                const double val = long_column_value(i, pos, deg);
                if constexpr (INPLACE) V[first_lcol + i] = val;
                if (!frozen && val < 0.0) {
                    const int32_t* const row = vr + (pos - vp);
                    for (int j = 0; j < deg; ++j) flip(row[j]);
                }
                for (int j = 0; j < deg; ++j) q_update(pos[j], val, Rload(pos[j]));
            }
            if (delta) atomicAdd(&unsat[p], delta);
            __syncthreads();                                          // ---- barrier B
            if (frozen) continue;                                     // forced mode after convergence
            const bool conv = unsat[p] == 0;                          // H hard == s
            if (conv || it == P.max_iter - 1) {
                emit(conv ? 1 : 0, it);
                frozen = true;
                if (!(conv && force_full) || it == P.max_iter - 1) break;
                __syncthreads();     // emission read R; the next check step overwrites it
            }
        }
        if (tid == 0) {
            if constexpr (!EARLY_FETCH)
                if (dynamic) fetched = atomicAdd(reinterpret_cast<unsigned*>(P.work_counter), 1u);
            *next_item = fetched;
        }
        __syncthreads();
        b = (long long)gridDim.x + (long long)*next_item;   // (next write: after the barrier at the loop top)
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
