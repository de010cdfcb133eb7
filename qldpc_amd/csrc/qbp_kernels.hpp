// Fused on-chip belief-propagation kernel for gfx950 (MI355X).
//
// One lane per CHECK node.  A workgroup decodes S syndromes concurrently ("slots"); slot s owns
// lanes [s*m, (s+1)*m).  A lane keeps its check's d_c variable->check messages Q in registers
// for the whole decode; the only data exchanged between lanes are the check->variable messages
// R, one double per edge, staged in LDS ([slot][edge j of the row][check c]: consecutive lanes
// write consecutive addresses).  Per BP iteration a lane
//   1. check step:   t_j = tanh(Q_j/2); row product; R_j = 2 atanh(clip(prod / t_j * sign))
//                    -> ds_write R_j                          (beliefPropagation.py:114-126)
//   2. barrier
//   3. variable step, done redundantly by every check lane for its own d_c variables: reads the
//      <= DV messages of each variable's column from LDS (ascending check order, the order
//      np.sum(R, axis=0) accumulates in), value_j = sum + prior_j, Q_j = value_j - R_j, hard
//      bit, parity of the row vs the syndrome bit -> per-slot "unsatisfied" flag (:129-139)
//   4. barrier, read the flag: converged / iteration limit -> emit outputs, fetch next syndrome.
//      (QBP_FLAG_FORCE_FULL launches of the (6, 3) shape skip this barrier: two copies of R, the flag
//      read one phase later -- see ONE_BAR in the kernel.)
// No message ever touches HBM: per syndrome the kernel reads m syndrome bytes and writes
// n hard bytes + n LLR doubles + 5 bytes.  Slots fetch work from a global atomic counter, so a
// slot whose syndrome converges early (reference semantics: return at the first syndrome match)
// immediately starts the next one, independent of its neighbours.
//
// Irregular rows/columns are padded: a missing edge has prior = +inf (its Q stays +inf, its tanh
// is exactly 1.0, its hard bit 0), a missing column entry points at a per-slot LDS word that
// holds 0.0 (x + 0.0 is exact) -- the kernel body has no degree-dependent branches.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "qbp_math.hpp"
#include "qbp_mc.hpp"

// Register-budget choices of the shapes that do not fit 128 registers otherwise (A/B switches for
// tools/build_variants.sh; the defaults are the measured-best settings, profiles/r02_ab_wide_mc.txt)
#ifndef QBP_WIDE_JG
#define QBP_WIDE_JG 4          // (8, 4) shape: edges per LDS gather group
#endif
#ifndef QBP_WIDE_HOLD_R
#define QBP_WIDE_HOLD_R 1      // (8, 4) shape: own messages in registers across barrier B1
#endif
#ifndef QBP_WIDE_PACK
#define QBP_WIDE_PACK 0        // (8, 4) shape: two LDS offsets per register
#endif
#ifndef QBP_SMALL_HOLD_R
#define QBP_SMALL_HOLD_R 1     // (6, 3) decode builds: as above
#endif
#ifndef QBP_SMALL_PACK
#define QBP_SMALL_PACK 0
#endif
#ifndef QBP_MC_HOLD_R
#define QBP_MC_HOLD_R 0        // Monte-Carlo builds: as above
#endif
#ifndef QBP_MC_PACK
#define QBP_MC_PACK 0
#endif
#ifndef QBP_WORK_CHUNK_SMALL_CODES
#define QBP_WORK_CHUNK_SMALL_CODES 1   // 0: at most eight syndromes per fetch whatever the code size (A/B)
#endif
#ifndef QBP_WORK_CHUNK_FIXED
#define QBP_WORK_CHUNK_FIXED 0    // 1: eight syndromes per fetch from the work counter, always (A/B)
#endif

namespace qbp {

// Forced-iteration launches with ONE workgroup barrier per iteration keep two copies of the
// check->variable messages in LDS; the second copy sits at this constant byte distance from the first
// (a constant, so that the toggle costs nothing: it lands in the offset field of the ds instructions of a
// loop unrolled by two).  The first copy, S * slot_stride doubles, must fit below it.
constexpr int FUSED_R2_OFF_BYTES = 57344;

struct FusedParams {
    // problem
    const uint8_t* syndromes;   // [B][m] (decode mode)
    const double* prior;        // [n]
    long long B;                // syndromes / trials in this launch
    int m, n;
    int S;                      // slots per workgroup
    int slot_stride;            // doubles per slot in LDS (DC*m + 2)
    int max_iter;
    unsigned flags;
    int padded;                 // some row has fewer than DC edges (irregular H)
    int n_words4;               // ceil(n / 4)
    int r0_table;               // LDS holds the first check step's messages (early-exit launches, when it fits)
    double alpha, damping, clip_llr;
    // outputs (decode mode; may be null)
    uint8_t* hard;
    uint8_t* converged;
    int32_t* iters;
    double* llr;
    // static tables of the code
    const int32_t* tab_var;     // [DC][m]   variable of edge j of check c, -1 = padding
    const uint16_t* tab_nbr;    // [DC][DV][m] LDS word offsets of the column of that variable
    const uint32_t* tab_writer; // [m] bit j: edge (c, j) is the first of its column
    const int32_t* iso_vars;    // variables with no check
    int n_iso;
    // work distribution
    unsigned long long* work_counter;   // zeroed before launch
    // Monte-Carlo mode
    const unsigned long long* lx_cols;  // [n] bit l = Lx[l][v]
    long long trial_begin;
    unsigned long long seed;
    unsigned threshold;         // floor(p * 2^32)
    int draws;
    int half_distance;          // distance // 2
    long long* counters;        // [NUM_COUNTERS], atomically added
    const uint8_t* errors_in;   // optional [B][n]: decode and classify THESE errors instead of sampling
    // Monte-Carlo + OSD: records of the trials BP did not converge on (indexed by the trial's
    // position b in this launch), and the list of those positions
    long long* fail_list;       // null = classify BP output directly
    unsigned long long* fail_count;
    uint8_t* fail_syn;          // [B][m]
    double* fail_llr;           // [B][n]
    uint8_t* fail_hard;         // [B][n]
    uint8_t* fail_err;          // [B][n]
};

// Rarely used launch parameters (output pointers, Monte-Carlo settings, ...) are re-read from the
// kernel-argument segment where they are needed instead of being kept in SGPRs for the whole
// kernel: the hot loop already needs ~60 SGPRs for FP64 constants, and every SGPR the compiler
// has to spill costs v_readlane/v_writelane slots on the (saturated) vector ALU.  The empty asm
// makes the pointer opaque so that the loads cannot be hoisted out of the cold branches.
typedef const FusedParams __attribute__((address_space(4)))* ColdArgs;
__device__ __forceinline__ ColdArgs cold_args()
{
    ColdArgs p = (ColdArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
#define COLD(field) (cold_args()->field)
// Markers around arithmetic that does not run every iteration, for the static instruction count of
// tools/valu_mix.py (no effect on the hardware beyond two one-cycle scalar no-ops inside that path).
#define QBP_COLD_BEGIN() asm volatile("s_nop 9")
#define QBP_COLD_END() asm volatile("s_nop 10")

// Syndromes a slot leader fetches at once when its syndromes have taken avg4 / 4 iterations on average
// (see "Work distribution" in the kernel).
__device__ __forceinline__ int work_chunk_for(int avg4)
{
    return avg4 < 12 ? 8 : avg4 < 28 ? 4 : avg4 < 60 ? 2 : 1;
}

__device__ __forceinline__ double clipd(double x, double lo, double hi)
{   // np.clip(x, lo, hi) == minimum(maximum(x, lo), hi) for non-NaN x: v_max_f64 + v_min_f64
    return __builtin_fmin(__builtin_fmax(x, lo), hi);
}

__device__ __forceinline__ double clipd_nan(double x, double lo, double hi)
{   // same, but a NaN stays a NaN as in np.clip (the damped variants can produce inf - inf when
    // a check has a single edge: rework/decoding.py keeps iterating on NaNs there)
    const double y = x < lo ? lo : x;
    return y > hi ? hi : y;
}

// Classification of one finished trial by its slot leader (paperResults_GPU.py:127-144 without
// the OSD call): reads and clears the slot's accumulators, bumps the slot's counter row in LDS.
__device__ __forceinline__ void mc_classify(unsigned long long* mc_lmask, int* mc_weight,
                                            int* mc_diff, int* cnt, int slot, int conv, int it,
                                            int half_distance, bool deferred)
{
    if (deferred) {            // handed to OSD: only the BP bookkeeping is counted here
        cnt[0] += 1; cnt[6] += 1; cnt[7] += it;
        return;
    }
    const unsigned long long lm = mc_lmask[slot];
    const int ew = mc_weight[slot];
    const int df = mc_diff[slot];
    mc_lmask[slot] = 0ull; mc_weight[slot] = 0; mc_diff[slot] = 0;
    mc_count_trial(cnt, lm, ew, df, conv, it, half_distance);
}

// LDS carve (in units of 8 bytes after the message area):
//   [DC][m] priors of the edges' variables (+inf for padding edges)
//   [S]  next work index per slot
//   [S]  end of the chunk of work indices the slot is drawing from (leader only)
//   [S]  MC logical-mask accumulator
//   then 32-bit words: flag[2][S], mc_weight[S], mc_diff[S], active_count,
//   mc_count[S][NUM_COUNTERS] (Monte-Carlo mode only), var_lds[DC][m], err_lds[2][S][n4] bytes (MC)
template <int DC, int DV, int VARIANT, bool MC, bool FORCE_FULL, int MAX_THREADS, int MIN_WAVES_PER_SIMD,
          bool ONE_BARRIER = false>
__global__ __launch_bounds__(MAX_THREADS, MIN_WAVES_PER_SIMD) void bp_fused_kernel(const FusedParams P)
{
    // ONE_BAR (forced-iteration decode launches only): one workgroup barrier per iteration instead of two.
    // The second barrier of an iteration orders (a) the next check step's writes of R after this variable
    // step's reads and (b) the slot's convergence flag before its readers.  (a) goes away with two copies
    // of R used alternately; (b) is deferred: whether iteration k satisfied the syndrome is looked at after
    // the barrier of iteration k + 1 -- the posterior values of iteration k are still in registers then --
    // which is only affordable when nothing is decided by it but WHEN the outputs are written: in forced
    // mode every syndrome runs max_iter iterations anyway (and all slots switch syndromes in the same
    // phase, so the last iteration, which must settle before the registers are reused, keeps its second
    // barrier for the whole workgroup).  With early exit the next check step would be wasted work.
    constexpr bool ONE_BAR = ONE_BARRIER && FORCE_FULL && !MC;
    // dynamic LDS: the tables of the two elementary functions (qbp_math.hpp, NpImage) first -- a constant
    // address, so that a table access is a row offset plus an immediate -- then the carve described above
    extern __shared__ __attribute__((aligned(16))) double smem_all[];
    constexpr NpT np_tab = 0u;          // = the LDS address of smem_all: this kernel has no static LDS (checked below)
    double* const smem = smem_all + NP_LDS_DOUBLES;
    const int tid = threadIdx.x;
    const int m = P.m;
    const int S = P.S;
    const int slot = tid / m;
    const int c = tid - slot * m;
    const bool lane_valid = slot < S;
    const bool leader = lane_valid && c == 0;
    const int sl = lane_valid ? slot : 0;

    double* const Rs = smem + (size_t)sl * P.slot_stride;
    const int zoff = DC * m;
    // priors of this check's variables, [edge j][check c], shared by all slots (an LDS read per
    // use instead of 12 VGPRs per lane for the whole kernel)
    double* const pri_lds = smem + (ONE_BAR ? FUSED_R2_OFF_BYTES / 8 : 0) + (size_t)S * P.slot_stride;
    // Early-exit launches: the check->variable messages of a syndrome's FIRST check step, which depend
    // on nothing but the priors and the check's syndrome bit -- r0_lds[bit][edge j][check c], computed once
    // per workgroup by the kernel's own check-step code (same bits).  At low error rates most syndromes
    // need one or two iterations, and the first check step is most of the first one.
    const bool use_r0 = !FORCE_FULL && P.r0_table != 0;
    double* const r0_lds = pri_lds + DC * m;
    long long* const next_work = reinterpret_cast<long long*>(r0_lds + (use_r0 ? 2 * DC * m : 0));
    // (the variable indices of the edges, needed only when a syndrome is emitted, sit behind the
    // 32-bit words below: var_lds[DC][m])
    long long* const chunk_ends = next_work + S;     // (LDS, not a register: touched once per syndrome)
    unsigned long long* const mc_lmask = reinterpret_cast<unsigned long long*>(chunk_ends + S);
    int* const words = reinterpret_cast<int*>(mc_lmask + S);
    int* const flag0 = words;            // [3][S] (two in use, three with ONE_BAR)
    int* const mc_weight = words + 3 * S;
    int* const mc_diff = words + 4 * S;
    int* const work_avg = words + 5 * S;     // [0]: 4 x running mean of the workgroup's iterations per syndrome ([S] reserved)
    int* const active_count = words + 6 * S;
    int* const mc_count = words + 6 * S + 1 + sl * NUM_COUNTERS;   // this slot's row
    int* const var_lds = words + 6 * S + 1 + S * NUM_COUNTERS;     // [DC][m], -1 = padding
    // Monte-Carlo mode: sampled error bytes of the slot's trials, [2][S][n4] (n4 = n rounded up to a
    // multiple of 4), written by the slot's first ceil(n/4) lanes one barrier before use.  Two
    // buffers per slot, used alternately by consecutive trials: the emission of a finished trial still
    // reads the bytes of its isolated variables after barrier B2, while faster waves of the same slot
    // may already be sampling the next trial at the top of the loop (no barrier in between).
    unsigned err_par = 0;                         // buffer of the trial being decoded (uniform per slot)
    // (recomputed where it is used -- all of them once-per-trial places -- instead of being held in
    // registers across the hot loop)
    auto err_buf = [&]() -> unsigned char* {
        const int n4 = COLD(n_words4) * 4;
        return reinterpret_cast<unsigned char*>(var_lds + DC * m) + (size_t)(sl + (err_par ? S : 0)) * n4;
    };

    // ---- check step of one row: q[DC] -> put(j, r) for its DC edges --------------------------------
    auto check_step = [&](const double (&q)[DC], unsigned sb, auto&& put) {
        if constexpr (VARIANT == 2) {
            // rework/decoding.py:28-56
            double sprod = 1.0, min1 = __builtin_inf();
            int min1_j = 0;
            bool anynan = false;
#pragma unroll
            for (int j = 0; j < DC; ++j) {
                const double s = q[j] < 0.0 ? -1.0 : 1.0;   // sign, 0 -> +1, padding +1
                sprod *= s;
                anynan |= q[j] != q[j];
                const double a = __builtin_fabs(q[j]);
                if (a < min1) { min1 = a; min1_j = j; }       // first occurrence
            }
            // np.sign(nan) = nan: one NaN message (inf - inf with infinite priors) makes the
            // row's sign product, hence every R of the row, NaN
            if (anynan) sprod = __builtin_nan("");
            double min2 = __builtin_inf();
#pragma unroll
            for (int j = 0; j < DC; ++j) {
                const double a = __builtin_fabs(q[j]);
                if (j != min1_j && a < min2) min2 = a;
            }
            const double as = sb ? -P.alpha : P.alpha;        // alpha * syndrome_sign
#pragma unroll
            for (int j = 0; j < DC; ++j) {
                const double s = q[j] < 0.0 ? -1.0 : 1.0;
                const double mag = (__builtin_fabs(q[j]) == min1) ? min2 : min1;
                put(j, (as * (sprod * s)) * mag);
            }
        } else {
            double t[DC];
            double prod;
#pragma unroll
            for (int j = 0; j < DC; ++j) {
                t[j] = tanh_half_msg<VARIANT>(q[j], np_tab);     // np.tanh(Q * 0.5), :114
                if constexpr (DC > 6) QBP_EDGE_FENCE();
            }
#pragma unroll
            for (int j = 0; j < DC; ++j) prod = (j == 0) ? t[0] : prod * t[j];     // np.prod, ascending column
            // t_safe = where(|t| < 1e-15, 1e-15, t) (:122).  |t| <= 1, so a row whose product is at least
            // 1e-15 in magnitude has no such factor: one wave-uniform test on the product replaces the six
            // compares and twelve selects in all but degenerate rows (messages of magnitude 1e-15, or six
            // messages near 0.006 at once -- those take the selects)
            if (__builtin_amdgcn_ballot_w64(!(__builtin_fabs(prod) >= 1e-15)) != 0ull) {
                // (rare path, bracketed for tools/valu_mix.py: the instruction count that prices the kernel must
                // not include it)
                QBP_COLD_BEGIN();
#pragma unroll
                for (int j = 0; j < DC; ++j) {
                    const double ts = __builtin_fabs(t[j]) < 1e-15 ? 1e-15 : t[j];
                    // (a zero or denormal product: quotients of any size, down to the subnormals)
                    const double r = check_message<VARIANT, false>(div_nr(prod, ts), sb, np_tab);
                    put(j, VARIANT == 1 ? r * P.alpha : r);
                }
                QBP_COLD_END();
            } else {
#pragma unroll
                for (int j = 0; j < DC; ++j) {
                    // prod / t, correctly rounded like numpy's division: 1e-15 <= |prod| <= 1 and
                    // 1e-15 <= |t| <= 1, so no operand scaling is needed (div_nr's precondition) and the
                    // quotient is at least 1e-15 in magnitude (check_message: NORMAL)
                    const double r = check_message<VARIANT, true>(div_nr(prod, t[j]), sb, np_tab);     // :123-126
                    put(j, VARIANT == 1 ? r * P.alpha : r);
                    if constexpr (DC > 6) QBP_EDGE_FENCE();
                }
            }
        }
    };

    // ---- per-lane static tables (registers for the whole kernel) ---------------------------
    // LDS word offsets of the columns of this check's variables.  The (6, 3) shape keeps one register
    // per offset; the (8, 4) shape and the Monte-Carlo builds pack two 16-bit offsets per register
    // (half the registers, one extra shift/mask per gather) -- their register budget is what limits them.
    constexpr bool PACK_NBR = (DC > 6 && QBP_WIDE_PACK != 0) || (MC && QBP_MC_PACK != 0) ||
                              (DC <= 6 && !MC && QBP_SMALL_PACK != 0);
    constexpr int NBR_W = PACK_NBR ? (DV + 1) / 2 : DV;
    unsigned nbr[DC][NBR_W];
    unsigned wmask = 0;
    unsigned vmask = 0;            // bit j: edge j of this check exists (not padding)
#pragma unroll
    for (int j = 0; j < DC; ++j) {
        const int v = lane_valid ? P.tab_var[j * m + c] : -1;
        vmask |= (v >= 0 ? 1u : 0u) << j;
        if (slot == 0) {
            pri_lds[j * m + c] = v >= 0 ? P.prior[v] : __builtin_inf();
            var_lds[j * m + c] = v;
        }
#pragma unroll
        for (int k = 0; k < DV; ++k)
        {
            const unsigned o = lane_valid ? P.tab_nbr[(j * DV + k) * m + c] : (unsigned)zoff;
            if constexpr (PACK_NBR) {
                if (k & 1) nbr[j][k / 2] |= o << 16; else nbr[j][k / 2] = o;
            } else {
                nbr[j][k] = o;      // (the compiler keeps the 18 byte addresses in registers)
            }
        }
    }
    if (lane_valid) wmask = P.tab_writer[c];
    if (lds_address(smem_all) != np_tab) __builtin_trap();
    np_tables_to_lds(smem_all, tid, blockDim.x);
    if (use_r0) __syncthreads();                    // (the table of first check steps uses them)
    if (use_r0 && lane_valid && slot == 0) {
        // (this lane wrote the priors it reads here; the table is published by the barrier before the loop)
        double qp[DC];
#pragma unroll
        for (int j = 0; j < DC; ++j) qp[j] = pri_lds[j * m + c];
        check_step(qp, 0u, [&](int j, double r) { r0_lds[j * m + c] = r; });
        check_step(qp, 1u, [&](int j, double r) { r0_lds[(DC + j) * m + c] = r; });
    }

    const long long B = P.B;
    const long long total_slots = (long long)gridDim.x * S;
    const double one_minus_damping = 1.0 - P.damping;
    const int max_iter = P.max_iter;

    // Work distribution: the first syndrome of a slot is static; afterwards the slot leader draws
    // chunks of consecutive indices from one global counter.  A single word sustains only ~88
    // dequeues/us (MI355X_MICROARCH.md 'dequeue'), which early-exit decoding at low error rates
    // would exceed with one atomic per syndrome -- and a chunk of eight syndromes that all run
    // max_iter iterations is a 8 x max_iter iteration tail on one slot while others idle (the
    // reference driver's 5 000-syndrome batches at p = 0.05, maxIter 150: 3.5 ms instead of 2.05).
    // Chunk = the larger of
    //   * a share of what is left: 8 / 4 / 2 / 1 syndromes while >= 16 / 8 / 4 / 0 per slot remain
    //     (judged by the slot's own last index: the counter may be a chunk per slot further), and
    //   * what the workgroup's recent syndromes cost: 8 while they took < 3 iterations on average,
    //     4 / 2 below 7 / 15, else 1 (work_chunk_for).  That running mean starts at max_iter, jumps
    //     up with one slow syndrome and decays by a quarter per finished one: the first syndromes
    //     to finish are the easy ones, and must not earn a slot eight hard ones.
    // Measured against a fixed chunk of 8 and three other rules: profiles/r02_ab_work_chunk.txt.
    auto work_chunk = [&](long long handed_out, int by_cost) -> int {
        if (QBP_WORK_CHUNK_FIXED) return FORCE_FULL ? 1 : 8;                       // (A/B: round 1's rule)
        const long long rem = B - handed_out;
        const int share = rem >= 16 * total_slots ? 8 : rem >= 8 * total_slots ? 4 : rem >= 4 * total_slots ? 2 : 1;
        const int ch = share > by_cost ? share : by_cost;
        // (small codes decode so many syndromes per microsecond -- [[72,12,6]]: 640 at p = 0.01 -- that eight per
        // atomic still is 80 fetches/us on one word: two or four times as many while plenty remain)
        const int f = QBP_WORK_CHUNK_SMALL_CODES ? (m <= 36 ? 4 : m <= 72 ? 2 : 1) : 1;
        return (f > 1 && ch == 8 && rem >= (long long)16 * f * total_slots) ? 8 * f : ch;
    };
    long long b = lane_valid ? (long long)blockIdx.x * S + slot : B;
    bool active = b < B;

    if (leader) {
        Rs[zoff] = 0.0;
        if constexpr (ONE_BAR) Rs[FUSED_R2_OFF_BYTES / 8 + zoff] = 0.0;
        flag0[slot] = 0;
        flag0[S + slot] = 0;
        flag0[2 * S + slot] = 0;
        mc_lmask[slot] = 0ull;
        mc_weight[slot] = 0;
        mc_diff[slot] = 0;
        if (slot == 0) *work_avg = 4 * P.max_iter;
        if constexpr (MC) {
            for (int i = 0; i < NUM_COUNTERS; ++i) mc_count[i] = 0;
        }
        // (a launch whose syndromes are all some slot's first one never touches the counter: tens of
        // thousands of atomics on one word in the same microsecond are not free -- 40 us for 10 000)
        const int ch = work_chunk(9 * total_slots, 1);      // (every slot fetches at this moment)
        const long long first =
            B > total_slots ? total_slots + (long long)atomicAdd(P.work_counter, (unsigned long long)ch) : B;
        next_work[slot] = first;
        chunk_ends[slot] = first + ch;
    }
    if (tid == 0) {
        long long first = (long long)blockIdx.x * S;
        long long cnt = B - first;
        *active_count = (int)(cnt < 0 ? 0 : (cnt > S ? S : cnt));
    }

    // The lane's own check->variable messages are needed again in the variable step (Q = value - R):
    // kept in registers across barrier B1, or re-read from LDS (a conflict-free ds_read per edge, no
    // vector-ALU cost) to save 2 * DC registers.  Measured (profiles/r02_ab_wide_mc.txt): the
    // Monte-Carlo builds are 6.6 % faster re-reading (their hot loop then runs without scratch
    // accesses); the (8, 4) shape is 4 % faster holding them, although that build spills 19
    // registers and the re-reading one none -- its iteration is not bound by those reloads.
    constexpr bool HOLD_R = MC ? (QBP_MC_HOLD_R != 0 && DC <= 6) : (DC <= 6 ? QBP_SMALL_HOLD_R != 0 : QBP_WIDE_HOLD_R != 0);

    // ---- per-syndrome state ---------------------------------------------------------------
    double Q[DC];
    double R[HOLD_R ? DC : 1];
    unsigned sbit = 0, ebits = 0;
    int it = 0;
    bool frozen = false;

    auto start_syndrome = [&]() {
        it = 0;
        frozen = false;
#pragma unroll
        for (int j = 0; j < DC; ++j) Q[j] = pri_lds[j * m + c];   // Q = where(mask, initialBelief, 0)
        if constexpr (MC) {
            ebits = 0;
            const unsigned char* const err_lds = err_buf();
#pragma unroll
            for (int j = 0; j < DC; ++j) {
                const int v = var_lds[j * m + c];
                if (v >= 0) ebits |= (unsigned)err_lds[v] << j;
            }
            sbit = __builtin_popcount(ebits) & 1u;       // syndrome = H e mod 2
        } else {
            sbit = COLD(syndromes)[b * m + c] & 1u;
        }
    };
    bool need_start = active;     // (re)initialise at the loop top, where Q / val / R are dead
    __syncthreads();

    // leader-only bookkeeping (refill: 0, or the size of the next chunk should one be needed)
    int refill = 0;
    bool mc_pending = false;
    int mc_pending_conv = 0, mc_pending_it = 0;

    double val_keep[DC];          // ONE_BAR: posterior values of the lane's edges, alive until the next phase
    // decode-mode emission of the values in val[] (one read of the output pointers per emission --
    // adjacent kernel arguments: a single scalar load -- not one per use)
    auto emit_decode = [&](const double (&val)[DC], bool conv, int it_out) {
        const ColdArgs ca = cold_args();
        double* const o_llr = ca->llr;
        uint8_t* const o_hard = ca->hard;
        const long long row = b * ca->n;
#pragma unroll
        for (int j = 0; j < DC; ++j) {
            if ((wmask >> j) & 1u) {
                const long long o = row + var_lds[j * m + c];
                if (o_llr) o_llr[o] = val[j];
                if (o_hard) o_hard[o] = (uint8_t)(val[j] < 0.0);
            }
        }
        const int n_iso = ca->n_iso;
        for (int i = c; i < n_iso; i += m) {
            const int v = COLD(iso_vars)[i];
            const double pv = COLD(prior)[v];
            if (o_llr) o_llr[row + v] = pv;
            if (o_hard) o_hard[row + v] = pv < 0.0;
        }
        if (c == 0) {
            if (ca->converged) ca->converged[b] = conv;
            if (ca->iters) ca->iters[b] = it_out;
        }
    };
    // flag buffers: written in phase p (f_cur), read after the next synchronisation, cleared by the leader
    // one phase ahead (f_next); ONE_BAR reads them one phase later (f_prev), hence three of them
    int f_cur = 0, f_next = 1, f_prev = 2;
    int wg_it = 0;                // ONE_BAR: iteration index of every active slot (they run in step)

    auto phase_body = [&](auto par_tag) -> bool {
        constexpr int PAR = decltype(par_tag)::value;
        double* const Rw = ONE_BAR ? Rs + PAR * (FUSED_R2_OFF_BYTES / 8) : Rs;     // this phase's copy of R
        double val_phase[DC];     // (two-barrier builds: the values do not outlive the phase)
        double (&val)[DC] = ONE_BAR ? val_keep : val_phase;
        if constexpr (MC) {
            // Sample the errors of a trial that starts in this phase: one Philox evaluation per
            // four qubits, by the first ceil(n/4) lanes of the slot; the extra barrier (Monte-Carlo
            // builds only) orders the bytes before the check lanes gather them.
            if (need_start) {
                err_par ^= 1u;
                unsigned* const err_words = reinterpret_cast<unsigned*>(err_buf());
                const unsigned long long trial = (unsigned long long)(COLD(trial_begin) + b);
                const uint8_t* const ein = COLD(errors_in);
                for (int g = c; g < P.n_words4; g += m)
                    err_words[g] = ein ? mc_stored_quad(ein + b * COLD(n), g, COLD(n))
                                       : mc_error_quad(trial, g, COLD(draws), COLD(seed), COLD(threshold));
            }
            __syncthreads();                                      // B0
        }
        if (need_start) { start_syndrome(); need_start = false; }
        // ================= check step =======================================================
        if (active) {
            if (use_r0 && it == 0) {
                // first check step of a syndrome: from the workgroup's table (see r0_lds)
                const double* const r0 = r0_lds + (sbit ? DC * m : 0);
#pragma unroll
                for (int j = 0; j < DC; ++j) {
                    const double r = r0[j * m + c];
                    Rw[j * m + c] = r;
                    if constexpr (HOLD_R) R[j] = r;
                }
            } else {
                check_step(Q, sbit, [&](int j, double r) {
                    Rw[j * m + c] = r;
                    if constexpr (HOLD_R) R[j] = r;
                });
            }
        }
        __syncthreads();                                          // B1
        if (*active_count == 0) return true;

        // ================= variable step (per edge of this check) ============================
        if constexpr (ONE_BAR) {
            // iteration it - 1 of this syndrome satisfied it (its flag is complete since the barrier above):
            // its posterior values, still in val[], are the outputs (beliefPropagationGPU.py:160-167)
            if (active && it > 0 && !frozen && flag0[f_prev * S + slot] == 0) {
                emit_decode(val, true, it - 1);
                frozen = true;
            }
        }
        if (active) {
            bool odd = sbit != 0;                                 // parity of the row vs syndrome
            // Issue the LDS gathers of a group of edges before the first add (one lgkmcnt wait per
            // group instead of two per edge): all 18 for the (6, 3) shape, two edges (8 + 2 reads) at
            // a time for the (8, 4) shape, whose register budget is the tighter constraint.
            constexpr int JG = (DC * DV <= 18) ? DC : QBP_WIDE_JG;
#pragma unroll
            for (int j0 = 0; j0 < DC; j0 += JG) {
                double rr[JG][DV];
                double rown[HOLD_R ? 1 : JG];
#pragma unroll
                for (int jj = 0; jj < JG; ++jj) {
#pragma unroll
                    for (int k = 0; k < DV; ++k)
                        if (j0 + jj < DC) {
                            if constexpr (PACK_NBR) {
                                unsigned o;
                                // (opaque: otherwise the 32 unpacked byte addresses are hoisted out
                                // of the iteration loop -- loop-invariant -- and spilled to scratch)
                                unsigned pk = nbr[j0 + jj][k / 2];
                                asm volatile("" : "+v"(pk));
                                o = (k & 1) ? pk >> 16 : pk & 0xffffu;
                                rr[jj][k] = Rw[o];
                            } else {
                                rr[jj][k] = Rw[nbr[j0 + jj][k]];
                            }
                        }
                    if constexpr (!HOLD_R)
                        if (j0 + jj < DC) rown[jj] = Rw[(j0 + jj) * m + c];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jj = 0; jj < JG; ++jj) {
                    const int j = j0 + jj;
                    if (j >= DC) break;
                    double s = rr[jj][0];
#pragma unroll
                    for (int k = 1; k < DV; ++k) s = s + rr[jj][k];   // ascending check order
                    val[j] = s + pri_lds[j * m + c];
                    odd ^= val[j] < 0.0;                              // hard decision: values < 0
                    double r_own;
                    if constexpr (HOLD_R) r_own = R[j]; else r_own = rown[jj];
                    const double qn = val[j] - r_own;
                    if constexpr (VARIANT == 0) {
                        Q[j] = qn;
                    } else {
                        const double q = P.damping * qn + one_minus_damping * Q[j];
                        Q[j] = clipd_nan(q, -P.clip_llr, P.clip_llr);
                    }
                }
            }
            if (P.padded) {                                       // wave-uniform branch
#pragma unroll
                for (int j = 0; j < DC; ++j)
                    if (!((vmask >> j) & 1u)) Q[j] = __builtin_inf();   // padding stays neutral
            }
            if (odd) flag0[f_cur * S + slot] = 1;                 // this check is unsatisfied
        }
        if (leader) {
            flag0[f_next * S + slot] = 0;
            if (refill) {
                long long nx = next_work[slot] + 1;
                if (nx == chunk_ends[slot]) {
                    const int ch = work_chunk(nx, refill);
                    nx = total_slots +
                         (long long)atomicAdd(COLD(work_counter), (unsigned long long)ch);
                    chunk_ends[slot] = nx + ch;
                }
                next_work[slot] = nx;
                refill = 0;
            }
            if constexpr (MC) {
                if (mc_pending) {
                    mc_classify(mc_lmask, mc_weight, mc_diff, mc_count, slot, mc_pending_conv,
                                mc_pending_it, COLD(half_distance),
                                COLD(fail_list) != nullptr && !mc_pending_conv);
                    mc_pending = false;
                }
            }
        }
        // (ONE_BAR: only the last iteration of a syndrome settles here; wg_it is workgroup-uniform)
        const bool settle = !ONE_BAR || wg_it == max_iter - 1;
        if (settle) __syncthreads();                              // B2

        // ================= convergence / output / next syndrome ==============================
        if (active && settle) {
            const bool conv = flag0[f_cur * S + slot] == 0;
            // (ONE_BAR: this block runs once per syndrome; reading the limit from the kernel-argument
            // segment here is also what tells tools/valu_mix.py so -- its rule for once-per-syndrome code)
            const bool last = it == (ONE_BAR ? COLD(max_iter) : max_iter) - 1;
            if (!frozen && (conv || last)) {
                if constexpr (MC) {
                    const ColdArgs ca = cold_args();     // one read of the cold arguments per emission
                    const long long row = b * ca->n;
                    const int n_iso = ca->n_iso;
                    if (ca->fail_list != nullptr && !conv) {
                        // BP failed: leave the trial to the OSD kernel (record indexed by b)
                        double* const f_llr = ca->fail_llr;
                        uint8_t* const f_hard = ca->fail_hard;
                        uint8_t* const f_err = ca->fail_err;
                        ca->fail_syn[b * m + c] = (uint8_t)sbit;
#pragma unroll
                        for (int j = 0; j < DC; ++j) {
                            if ((wmask >> j) & 1u) {
                                const long long o = row + var_lds[j * m + c];
                                f_llr[o] = val[j];
                                f_hard[o] = (uint8_t)(val[j] < 0.0);
                                f_err[o] = (uint8_t)((ebits >> j) & 1u);
                            }
                        }
                        const unsigned char* const err_lds = err_buf();
                        for (int i = c; i < n_iso; i += m) {
                            const int v = COLD(iso_vars)[i];
                            const double pv = COLD(prior)[v];
                            f_llr[row + v] = pv;
                            f_hard[row + v] = (uint8_t)(pv < 0.0);
                            f_err[row + v] = err_lds[v];
                        }
                        if (c == 0) ca->fail_list[atomicAdd(ca->fail_count, 1ull)] = b;
                    } else {
                    unsigned long long lm = 0ull;
                    int ew = 0;
                    unsigned df = 0;
                    const unsigned long long* const lx = ca->lx_cols;
#pragma unroll
                    for (int j = 0; j < DC; ++j) {
                        if ((wmask >> j) & 1u) {
                            const unsigned e = (ebits >> j) & 1u;
                            const unsigned res = (val[j] < 0.0 ? 1u : 0u) ^ e;
                            ew += (int)e;
                            df |= res;
                            const int v = var_lds[j * m + c];
                            if (res) lm ^= lx[v];
                        }
                    }
                    const unsigned char* const err_lds = err_buf();
                    for (int i = c; i < n_iso; i += m) {
                        const int v = COLD(iso_vars)[i];
                        const unsigned e = err_lds[v];
                        const unsigned res = (COLD(prior)[v] < 0.0 ? 1u : 0u) ^ e;
                        ew += (int)e;
                        df |= res;
                        if (res) lm ^= lx[v];
                    }
                    if (lm) atomicXor(&mc_lmask[slot], lm);
                    if (ew) atomicAdd(&mc_weight[slot], ew);
                    if (df) atomicOr(&mc_diff[slot], 1);
                    }
                    if (c == 0) { mc_pending = true; mc_pending_conv = conv; mc_pending_it = it; }
                } else {
                    emit_decode(val, conv, it);
                }
            }
            if (conv) frozen = true;
            const bool finished = last || (conv && !FORCE_FULL);
            if (finished) {
                b = next_work[slot];
                if (b < B) {
                    need_start = true;
                    if (c == 0) {                  // (the counter only grows: past B, nothing to fetch)
                        if constexpr (FORCE_FULL) {
                            refill = 1;
                        } else {
                            // (shared by the slot leaders of the workgroup: a lost update is harmless)
                            const int a4 = *work_avg;
                            const int n4 = it * 4 > a4 ? it * 4 : a4 - (a4 >> 2) + it;
                            *work_avg = n4;
                            refill = work_chunk_for(n4);
                        }
                    }
                } else {
                    active = false;
                    if (c == 0) atomicSub(active_count, 1);
                }
            } else {
                ++it;
            }
        } else if (active) {
            ++it;                                                 // (ONE_BAR, not the last iteration)
        }
        {   // next phase
            const int t = f_prev; f_prev = f_cur; f_cur = f_next; f_next = ONE_BAR ? t : f_prev;
            if constexpr (ONE_BAR) wg_it = wg_it == max_iter - 1 ? 0 : wg_it + 1;
        }
        return false;
    };
    for (;;) {
        if (phase_body(std::integral_constant<int, 0>{})) break;
        if constexpr (ONE_BAR) {
            if (phase_body(std::integral_constant<int, 1>{})) break;
        }
    }

    if constexpr (MC) {
        if (leader) {
            // the last emitted trial of this slot may still be pending (emitted after B2 of the
            // final phase; everybody passed B1 since, so the accumulators are complete)
            if (mc_pending)
                mc_classify(mc_lmask, mc_weight, mc_diff, mc_count, slot, mc_pending_conv,
                            mc_pending_it, COLD(half_distance),
                            COLD(fail_list) != nullptr && !mc_pending_conv);
            for (int i = 0; i < NUM_COUNTERS; ++i)
                if (mc_count[i])
                    atomicAdd(reinterpret_cast<unsigned long long*>(COLD(counters) + i),
                              (unsigned long long)mc_count[i]);
        }
    }
}

#ifdef QBP_DEFINE_KERNELS   /* non-template kernels: defined in their translation unit only */
// Device evaluation of the math functions (accuracy tests).
__global__ void debug_math_kernel(int kind, const double* x, double* y, long long n)
{
    __shared__ __attribute__((aligned(16))) double np_lds[NP_LDS_DOUBLES];
    np_tables_to_lds(np_lds, threadIdx.x, blockDim.x);
    const NpT np_tab = lds_address(np_lds);
    __syncthreads();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    switch (kind) {
        case 0: y[i] = tanh_half_msg<0>(v, np_tab); break;        // the kernels' tanh(q / 2)
        case 1: y[i] = atanh2_msg(v, np_tab); break;               // the kernels' 2 atanh(y)
        case 4: y[i] = tanh_half(v); break;                        // round-1/2 forms (QBP_MATH_FAST builds)
        case 5: y[i] = atanh2(v); break;
        case 2: y[i] = __builtin_amdgcn_rcp(v); break;          // raw v_rcp_f64 (seed accuracy)
        default: y[i] = div_nr(1.0, v); break;
    }
}

// The Monte-Carlo sampler on its own (qbp_mc_sample_errors): errors [T][n] of trials trial_begin .. + T, one
// Philox evaluation per four qubits -- the bytes every Monte-Carlo kernel draws for itself (mc_error_quad).
__global__ void mc_sample_kernel(uint8_t* errors, int n, long long T, long long trial_begin, int draws,
                                 unsigned long long seed, unsigned threshold)
{
    const int n4 = (n + 3) / 4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * n4) return;
    const long long b = i / n4;
    const int g = (int)(i - b * n4);
    const unsigned q = mc_error_quad((unsigned long long)(trial_begin + b), g, draws, seed, threshold);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (4 * g + k < n) errors[b * n + 4 * g + k] = (uint8_t)((q >> (8 * k)) & 1u);
}

#endif  // QBP_DEFINE_KERNELS

}  // namespace qbp
