// OSD-0 post-processing on the GPU: decoding/OSD.py:3-72 (performOSD + gf2_elimination).
//
// One wavefront per syndrome.  The reference permutes the columns of H by ascending |LLR|, runs a
// Gauss-Jordan elimination with row swaps and reads the solution off the pivot columns.  For a syndrome in
// the column space of H (every syndrome that comes from an error) the solution only depends on WHICH columns
// become pivots (the greedy, first-independent-columns basis in reliability order) -- not on which row
// serves as the pivot (for the others see osd_flag_inconsistent below) -- so this kernel never
// swaps or permutes: it keeps the bit-packed rows of H in LDS in their original column indexing,
// walks the columns in sorted order, picks any not-yet-used row with a 1 in that column as the
// pivot and XORs it into every other row that has the bit (rows are n bits + the syndrome bit).
// Sorting: bitonic network over (|llr| as its monotone bit pattern, column index) pairs in LDS; ties
// in |llr| fall back to the column index (the reference's np.argsort is unstable there;
// oracle/bp_oracle.c does the same); NaN sorts last, as in numpy.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qbp {

struct OsdParams {
    // code
    int m, n, W;                    // W = 32-bit words per row of H ((n + 31) / 32)
    int NP;                         // power of two >= n (sort width)
    int rank;                       // rank of H over GF(2): no pivot exists beyond it
    const uint32_t* hbits;          // [m][W] bit-packed rows of H
    const int32_t* row_ptr;         // CSR of H
    const int32_t* col_idx;
    // batch: record index of syndrome i is list ? list[i] : i
    long long count;
    const long long* count_ptr;     // optional: number of records (device), overrides count
    const long long* list;
    const uint8_t* syndromes;       // [*][m]
    const double* llr;              // [*][n]
    const uint8_t* hard;            // [*][n]
    uint8_t* solution;              // [*][n] (may be null in Monte-Carlo mode)
    long long* redo;                // optional: [0] counts, [1..] lists the records whose sweep ended with a
                                    // syndrome bit left on a row without a pivot (see osd_flag_inconsistent)
    // Monte-Carlo classification (paperResults_GPU.py:127-144 on the OSD output)
    const uint8_t* errors;          // [*][n] or null
    const unsigned long long* lx_cols;
    int half_distance;
    long long* counters;
};

// Sort key of |llr|: the IEEE bit pattern of a non-negative double is monotone as an unsigned
// integer (finite < inf < NaN), which is also numpy's order (np.argsort puts NaN last); every NaN
// gets the same pattern so that NaNs, like other ties, are ordered by column index.  A total order:
// the sorting network stays a permutation of the real columns whatever the LLRs contain.
__device__ __forceinline__ unsigned long long osd_order_key(double llr)
{
    const double a = __builtin_fabs(llr);
    return a != a ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(a);
}

__device__ __forceinline__ bool osd_less(unsigned long long ka, int ia, unsigned long long kb, int ib)
{
    return ka < kb || (ka == kb && ia < ib);
}

// LDS: { u64 keys[NP] | uint32 A[m][W+1] } (the sort is over before the matrix is filled: one region
// for both -- 8.7 instead of 12.8 KB for [[288,12,18]], 18 instead of 12 resident wavefronts per CU for
// a kernel that lives on hiding LDS latency); uint16 idx[NP]; int pivcol[m]; uint8 sol[n]
// (osd_lds_bytes).  WW = W + 1 at compile time (0: any width, at most 32 rows per lane).
__host__ __device__ inline size_t osd_region0_bytes(int m, int W, int NP)
{
    const size_t a = (size_t)NP * 8, b = (size_t)m * (W + 1) * 4;
    return ((a > b ? a : b) + 7) & ~(size_t)7;
}
__host__ __device__ inline size_t osd_lds_bytes(int m, int n, int W, int NP)
{
    return osd_region0_bytes(m, W, NP) + (size_t)NP * 2 + (size_t)m * 4 + (size_t)n + 16;
}

// A syndrome outside the column space of H (no error produces one; no caller of the reference passes one) ends its
// sweep with a 1 left in the syndrome column of a row without a pivot.  There, and only there, the reference's
// output depends on WHICH row served as the pivot of a column -- on its row swaps (OSD.py:56-59).  The fast kernels
// pick the first unused row in the original order; they list such a record in P.redo, and the host launches
// osd0_big_kernel, which follows the swaps, on the list (normally empty: the launch reads one counter and ends).
__device__ __forceinline__ void osd_flag_inconsistent(const OsdParams& P, long long rec)
{
    if (P.redo) {
        const unsigned long long at = atomicAdd(reinterpret_cast<unsigned long long*>(P.redo), 1ull);
        P.redo[1 + at] = rec;
    }
}

template <int WW>
__global__ __launch_bounds__(64) void osd0_kernel(const OsdParams P)
{
    extern __shared__ double osd_smem[];
    const int lane = threadIdx.x;
    // (row stride W + 1 words.  An odd stride -- 11 for n = 288 -- was measured: no gain, the two halves of a
    // wavefront are served separately and lanes l, l + 32 are the only ones an even stride maps to one bank)
    // (the row stride as a compile-time constant where the instantiation knows it: the ten accesses of a
    // row become wide LDS instructions)
    const int m = P.m, n = P.n, W = WW > 0 ? WW - 1 : P.W, NP = P.NP, RS = W + 1;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(osd_smem);
    uint32_t* A = reinterpret_cast<uint32_t*>(osd_smem);             // (after the sort: same bytes)
    uint16_t* idx = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(osd_smem) + osd_region0_bytes(m, W, NP));
    int* pivcol = reinterpret_cast<int*>(idx + NP);
    uint8_t* sol = reinterpret_cast<uint8_t*>(pivcol + m);

    const long long total = P.count_ptr ? *P.count_ptr : P.count;
    for (long long item = blockIdx.x; item < total; item += gridDim.x) {
        const long long rec = P.list ? P.list[item] : item;
        const double* llr = P.llr + rec * n;
        const uint8_t* hard = P.hard + rec * n;
        const uint8_t* syn = P.syndromes + rec * m;

        // ---- 1. ordering = argsort(|llr|)                                    OSD.py:10-11
        for (int i = lane; i < NP; i += 64) {
            keys[i] = i < n ? osd_order_key(llr[i]) : ~0ull;      // padding sorts behind everything
            idx[i] = (uint16_t)i;
        }
        for (int i = lane; i < n; i += 64) sol[i] = hard[i] & 1u;
        __syncthreads();
        for (int k = 2; k <= NP; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = lane; t < NP / 2; t += 64) {
                    const int lo = ((t / j) * (2 * j)) + (t % j);   // element with bit j clear
                    const int hi = lo + j;
                    const bool up = (lo & k) == 0;
                    const unsigned long long ka = keys[lo], kb = keys[hi];
                    const int ia = idx[lo], ib = idx[hi];
                    if (osd_less(kb, ib, ka, ia) == up) {
                        keys[lo] = kb; keys[hi] = ka; idx[lo] = (uint16_t)ib; idx[hi] = (uint16_t)ia;
                    }
                }
                __syncthreads();
            }
        }
        // ---- 2. A = [H | residual syndrome], residual = syndrome + hard @ H.T  OSD.py:7-8
        // (A overwrites the sort keys: every lane passed the sort's last barrier)
        unsigned sb = 0;                             // bit i: reduced syndrome bit of row lane + 64 i
        for (int r = lane, i = 0; r < m; r += 64, ++i) {
            for (int w = 0; w < W; ++w) A[r * RS + w] = P.hbits[r * W + w];
            unsigned par = syn[r] & 1u;
            for (int e = P.row_ptr[r]; e < P.row_ptr[r + 1]; ++e) par ^= sol[P.col_idx[e]];
            A[r * RS + W] = par;
            sb |= par << i;
            pivcol[r] = -1;
        }
        __syncthreads();
        // ---- 3. Gauss-Jordan over the columns in reliability order            OSD.py:31-72
        // Lane l owns rows l, l + 64, ...: per column one LDS read per owned row decides both the
        // pivot search (ballot) and which rows take the XOR; the pivot row is read once (broadcast)
        // into registers when it fits (WW = words per row incl. the syndrome word, compile time).
        int rank = 0;
        unsigned used = 0;                           // bit i: row lane + 64 i already serves as a pivot row
        // The sweep ends at the rank of H (:42-43 stops at m rows; rank(H) <= m) -- or as soon as no row without
        // a pivot has a 1 left in the syndrome column: every pivot found from there on would be chosen with a
        // reduced syndrome bit of 0, XOR nothing into the syndrome bits of the rows above it and put a 0 at its
        // own column, so the solution is already what the full sweep leaves.  (The residual syndrome of a BP
        // failure is light: the sweep typically ends after a small part of the columns.)
        for (int k = 0; k < n && rank < P.rank; ++k) {
            if (!__ballot((sb & ~used) != 0u)) break;
            const int c = idx[k];
            const int wi = c >> 5;
            const uint32_t bit = 1u << (c & 31);
            int p = -1;
            unsigned has = 0;                        // bit i: row lane + 64 i has a 1 in column c
            for (int base = 0, i = 0; base < m; base += 64, ++i) {
                const int r = base + lane;
                const bool one = r < m && (A[r * RS + wi] & bit);
                has |= (one ? 1u : 0u) << i;
                const unsigned long long mask = __ballot(one && !((used >> i) & 1u));
                if (p < 0 && mask) p = base + (int)__builtin_ctzll(mask);
            }
            if (p < 0) continue;                     // column depends on earlier ones (:52-53)
            ++rank;
            if (lane == (p & 63)) used |= 1u << (p >> 6);
            if constexpr (WW > 0) {
                uint32_t prow[WW];
#pragma unroll
                for (int w = 0; w < WW; ++w) prow[w] = A[p * RS + w];
                for (int r = lane, i = 0; r < m; r += 64, ++i) {
                    if (r != p && ((has >> i) & 1u)) {
#pragma unroll
                        for (int w = 0; w < WW; ++w) A[r * RS + w] ^= prow[w];              // :63-68
                        sb ^= (prow[WW - 1] & 1u) << i;
                    }
                }
            } else {
                const unsigned ps = A[p * RS + W] & 1u;
                for (int r = lane, i = 0; r < m; r += 64, ++i) {
                    if (r != p && ((has >> i) & 1u)) {
                        for (int w = 0; w <= W; ++w) A[r * RS + w] ^= A[p * RS + w];        // :63-68
                        sb ^= ps << i;
                    }
                }
            }
            if (lane == 0) pivcol[p] = c;
            __syncthreads();
        }
        if (__ballot((sb & ~used) != 0u) && lane == 0) osd_flag_inconsistent(P, rec);
        // ---- 4. e[pivot column] = reduced syndrome bit; solution = hard + e    OSD.py:14-26
        for (int r = lane; r < m; r += 64) {
            const int c = pivcol[r];
            if (c >= 0 && (A[r * RS + W] & 1u)) sol[c] ^= 1u;
        }
        __syncthreads();
        if (P.solution)
            for (int i = lane; i < n; i += 64) P.solution[rec * n + i] = sol[i];

        if (P.errors) {
            // classification of the OSD output (paperResults_GPU.py:127-144)
            const uint8_t* err = P.errors + rec * n;
            unsigned long long lm = 0ull;
            int ew = 0;
            unsigned df = 0;
            for (int i = lane; i < n; i += 64) {
                const unsigned e = err[i] & 1u;
                const unsigned res = sol[i] ^ e;
                ew += (int)e;
                df |= res;
                if (res) lm ^= P.lx_cols[i];
            }
            unsigned bad = 0;                        // is_valid_osd: (detection @ H.T) % 2 == syndrome
            for (int r = lane; r < m; r += 64) {
                unsigned par = syn[r] & 1u;
                for (int e = P.row_ptr[r]; e < P.row_ptr[r + 1]; ++e) par ^= sol[P.col_idx[e]];
                bad |= par;
            }
            for (int off = 32; off > 0; off >>= 1) {
                lm ^= __shfl_xor(lm, off);
                ew += __shfl_xor(ew, off);
                df |= __shfl_xor(df, off);
                bad |= __shfl_xor(bad, off);
            }
            if (lane == 0) {
                auto add = [&](int i) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + i), 1ull);
                };
                const bool logical = lm != 0ull;
                if (!bad && !logical && df) add(5);
                if (logical) {
                    add(1);
                    add(ew < P.half_distance ? 3 : 4);
                    add(8);
                }
                if (!df) add(9);
                if (bad) add(10);                    // OSD output that misses the syndrome (never)
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// OSD-0 for matrices whose bit-packed rows do not fit the LDS of one wavefront's workgroup (space-time
// and circuit-level matrices: 2592 x 7776 is 2.5 MB per syndrome).  One workgroup per syndrome; the
// working copy of [H | residual syndrome] lives in a per-workgroup global workspace, stored TRANSPOSED
// (word w of row r at At[w * m + r]) so that the threads of a wavefront -- one matrix row each -- read
// and write consecutive addresses; sort keys in LDS when n <= 8192, else in the workspace too.
// Same algorithm and tie rules as osd0_kernel above, hence the same solutions (tested on the small
// codes, where both kernels apply) -- one pivot at a time, full-width rows, and the one kernel that follows the
// reference's row swaps: it serves matrices beyond 8192 rows and the records the fast kernels list in P.redo.
struct OsdBigWorkspace {
    uint32_t* At;                   // [grid][(W + 1) * m]
    int32_t* pivcol;                // [grid][m]
    int32_t* posn;                  // [grid][2 m]: position of every row in the reference's swapped arrangement,
                                    // then the row at every position (osd0_big_kernel)
    uint8_t* sol;                   // [grid][n]
    unsigned long long* keys;       // [grid][NP]   (only when the keys do not fit LDS)
    int32_t* idx;                   // [grid][NP]
    int keys_in_lds;
    // osd0_blocked_kernel only (At: 64-bit words there, [grid][wc_max * m])
    int wc_max;                     // word planes of a full-width working copy: (n + 63) / 64 + 1
    int k_first;                    // sorted columns the first sweep keeps
    int lds_region0, lds_table;     // bytes: {keys | pos | table} region; the part of it the table may use
    int lds_act;                    // byte offset of the list of rows a block updates (m words)
    unsigned long long* next;       // work counter (zero at launch): syndromes are handed out one at a time
};

#ifdef QBP_DEFINE_KERNELS   /* non-template kernel: defined in its translation unit only */
__global__ __launch_bounds__(256) void osd0_big_kernel(const OsdParams P, const OsdBigWorkspace Wk)
{
    extern __shared__ double osd_smem[];
    __shared__ unsigned long long s_piv[2];   // (two slots, by column parity: see the pivot search)
    __shared__ unsigned long long s_lm;
    __shared__ int s_ew, s_df, s_bad;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int m = P.m, n = P.n, W = P.W, NP = P.NP, RS = P.W + 1;
    unsigned long long* keys;
    int* idx;
    if (Wk.keys_in_lds) {
        keys = reinterpret_cast<unsigned long long*>(osd_smem);
        idx = reinterpret_cast<int*>(keys + NP);
    } else {
        keys = Wk.keys + (size_t)blockIdx.x * NP;
        idx = Wk.idx + (size_t)blockIdx.x * NP;
    }
    uint32_t* const At = Wk.At + (size_t)blockIdx.x * RS * m;
    int* const pivcol = Wk.pivcol + (size_t)blockIdx.x * m;
    int* const posn = Wk.posn + (size_t)blockIdx.x * 2 * m;        // position of row r in the swapped arrangement
    int* const rowat = posn + m;                                   // row at position q
    uint8_t* const sol = Wk.sol + (size_t)blockIdx.x * n;

    const long long total = P.count_ptr ? *P.count_ptr : P.count;
    for (long long item = blockIdx.x; item < total; item += gridDim.x) {
        const long long rec = P.list ? P.list[item] : item;
        const double* llr = P.llr + rec * n;
        const uint8_t* hard = P.hard + rec * n;
        const uint8_t* syn = P.syndromes + rec * m;
        // ---- 1. ordering = argsort(|llr|), ties by column index                    OSD.py:10-11
        for (int i = tid; i < NP; i += nt) {
            keys[i] = i < n ? osd_order_key(llr[i]) : ~0ull;
            idx[i] = i;
        }
        for (int i = tid; i < n; i += nt) sol[i] = hard[i] & 1u;
        __syncthreads();
        for (int k = 2; k <= NP; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < NP / 2; t += nt) {
                    const int lo = ((t / j) * (2 * j)) + (t % j);
                    const int hi = lo + j;
                    const bool up = (lo & k) == 0;
                    const unsigned long long ka = keys[lo], kb = keys[hi];
                    const int ia = idx[lo], ib = idx[hi];
                    if (osd_less(kb, ib, ka, ia) == up) {
                        keys[lo] = kb; keys[hi] = ka; idx[lo] = ib; idx[hi] = ia;
                    }
                }
                __syncthreads();
            }
        }
        // ---- 2. A = [H | residual syndrome]                                         OSD.py:7-8
        for (int r = tid; r < m; r += nt) {
            for (int w = 0; w < W; ++w) At[(size_t)w * m + r] = P.hbits[(size_t)r * W + w];
            unsigned par = syn[r] & 1u;
            for (int e = P.row_ptr[r]; e < P.row_ptr[r + 1]; ++e) par ^= sol[P.col_idx[e]];
            At[(size_t)W * m + r] = par;
            pivcol[r] = -1;
            posn[r] = r;
            rowat[r] = r;
        }
        __syncthreads();
        // ---- 3. Gauss-Jordan over the columns in reliability order                  OSD.py:31-72
        // The pivot of a column is the first row at or below position `rank` of the reference's arrangement,
        // which swaps every pivot row up (:46-59): rows keep their place here and carry that position along.
        int rank = 0;
        for (int k = 0; k < n && rank < P.rank; ++k) {
            const int c = idx[k];
            const uint32_t* const colw = At + (size_t)(c >> 5) * m;
            const uint32_t bit = 1u << (c & 31);
            // (slot k & 1: a column without pivot leaves this iteration without a trailing barrier, so the
            // next column's reset must not touch the word the slower threads are still reading)
            unsigned long long* const piv = &s_piv[k & 1];
            if (tid == 0) *piv = ~0ull;
            __syncthreads();
            unsigned long long mine = ~0ull;          // (position, row)
            for (int r = tid; r < m; r += nt)
                if ((colw[r] & bit) && pivcol[r] < 0) {
                    const unsigned long long key = ((unsigned long long)(unsigned)posn[r] << 32) | (unsigned)r;
                    mine = key < mine ? key : mine;
                }
            if (mine != ~0ull) atomicMin(piv, mine);
            __syncthreads();
            const unsigned long long key = *piv;      // the first unused row with a 1 (:46-50)
            if (key == ~0ull) continue;               // column depends on earlier ones (:52-53)
            const int q = (int)(key >> 32), p = (int)(unsigned)key;
            if (tid == 0 && q != rank) {              // :56-59 (only this thread reads rowat; the trailing barrier
                                                      // publishes posn)
                const int r0 = rowat[rank];
                rowat[rank] = p; rowat[q] = r0;
                posn[p] = rank; posn[r0] = q;
            }
            ++rank;
            for (int r = tid; r < m; r += nt) {
                if (r != p && (colw[r] & bit)) {
                    // (the word holding bit c goes last: it is the loop's own condition for no one,
                    // but rows are independent, so any order is fine; keep it simple)
                    for (int w = 0; w <= W; ++w) At[(size_t)w * m + r] ^= At[(size_t)w * m + p];   // :63-68
                }
            }
            if (tid == 0) pivcol[p] = c;
            __syncthreads();
        }
        // ---- 4. e[pivot column] = reduced syndrome bit; solution = hard + e         OSD.py:14-26
        for (int r = tid; r < m; r += nt) {
            const int c = pivcol[r];
            if (c >= 0 && (At[(size_t)W * m + r] & 1u)) sol[c] ^= 1u;   // distinct pivot columns: no race
        }
        if (tid == 0) { s_lm = 0ull; s_ew = 0; s_df = 0; s_bad = 0; }
        __syncthreads();
        if (P.solution)
            for (int i = tid; i < n; i += nt) P.solution[rec * n + i] = sol[i];
        if (P.errors) {
            const uint8_t* err = P.errors + rec * n;
            unsigned long long lm = 0ull;
            int ew = 0;
            unsigned df = 0, bad = 0;
            for (int i = tid; i < n; i += nt) {
                const unsigned e = err[i] & 1u;
                const unsigned res = sol[i] ^ e;
                ew += (int)e;
                df |= res;
                if (res) lm ^= P.lx_cols[i];
            }
            for (int r = tid; r < m; r += nt) {
                unsigned par = syn[r] & 1u;
                for (int e = P.row_ptr[r]; e < P.row_ptr[r + 1]; ++e) par ^= sol[P.col_idx[e]];
                bad |= par;
            }
            if (lm) atomicXor(&s_lm, lm);
            if (ew) atomicAdd(&s_ew, ew);
            if (df) atomicOr(&s_df, 1);
            if (bad) atomicOr(&s_bad, 1);
            __syncthreads();
            if (tid == 0) {
                auto add = [&](int i) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + i), 1ull);
                };
                const bool logical = s_lm != 0ull;
                if (!s_bad && !logical && s_df) add(5);
                if (logical) {
                    add(1);
                    add(s_ew < P.half_distance ? 3 : 4);
                    add(8);
                }
                if (!s_df) add(9);
                if (s_bad) add(10);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// The same OSD-0, eight pivots at a time ("four Russians" blocking of the Gauss-Jordan sweep): the kernel
// for space-time / circuit-level matrices of up to 8192 rows.  One workgroup of 1024 threads per syndrome.
//
//  * A sweep ends as soon as no row without a pivot has a syndrome bit left (see osd0_kernel): after some 200
//    of the 5184 / 7776 columns on the BP failures of the tests' space-time matrices, 1632 at most.  So only
//    the first K = 1024 columns in reliability order are kept: the working copy holds H[:, order[:K]] in
//    SORTED column order, built from the CSR through the inverse permutation: 64-bit words, transposed
//    (word w of row r at At[w * m + r]: a wavefront's rows are consecutive), the residual syndrome bit in a
//    word plane of its own behind them.  A sweep that runs out of columns before it may end starts over with
//    4 K columns, then 16 K, ... n (tests force that path: QBP_OPT_OSD_BIG = 3 begins with K = 24).
//  * Sorted order means column k is never looked at again once it is passed: row operations only touch the
//    word planes from k / 64 on.
//  * A block is T <= 8 neighbouring columns inside one word.  Every thread keeps the T bits of its rows in a
//    register (a "byte"), and the sweep over the block's columns works on those bytes alone: per column one
//    LDS atomicMin (first not-yet-used row with the bit: decoding/OSD.py:46-50) and ONE barrier; a row that
//    has the bit XORs the pivot's byte into its own and notes which of the block's pivot rows it has to take
//    (D, a T-bit set over the pivot rows AS THEY WERE WHEN THE BLOCK BEGAN; a pivot row's own D at the moment
//    it is chosen travels with it through the atomicMin word).  Then the block's pivot rows go to LDS, all
//    2^T XOR combinations of them are tabulated there, and every row of the matrix is updated ONCE:
//    row ^= table[D].  Global traffic per block: one read-modify-write of the live word planes instead of
//    one per pivot.
//  * Same pivot columns as the row-at-a-time kernels (the greedy basis in reliability order does not depend
//    on the blocking), hence the same solutions: tests/test_gpu_osd.py compares all three kernels.
#ifdef QBP_OSD_TIMING   /* tools/osd_phases.py: cycles of thread 0 per phase, summed over workgroups */
__device__ unsigned long long g_osd_timing[8];
#define OSD_T0() long long t_prev = clock64(); unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define OSD_T(i) do { const long long t_now = clock64(); t_acc[i] += (unsigned long long)(t_now - t_prev); t_prev = t_now; } while (0)
#define OSD_TEND() do { if (threadIdx.x == 0) for (int q = 0; q < 8; ++q) atomicAdd(&g_osd_timing[q], t_acc[q]); } while (0)
__device__ unsigned long long g_osd_stat[8];   /* blocks, active rows, live words of active rows, columns swept */
#define OSD_STAT(i, v) atomicAdd(&g_osd_stat[i], (unsigned long long)(v))
#define OSD_SYNC() __syncthreads()
#else
#define OSD_STAT(i, v) do {} while (0)
#define OSD_SYNC() do {} while (0)
#define OSD_T0() do {} while (0)
#define OSD_T(i) do {} while (0)
#define OSD_TEND() do {} while (0)
#endif

template <int RPT>
__global__ __launch_bounds__(1024) void osd0_blocked_kernel(const OsdParams P, const OsdBigWorkspace Wk)
{
    extern __shared__ double osd_smem[];
    __shared__ unsigned s_piv[3];
    __shared__ unsigned s_nact[2];
    __shared__ unsigned long long s_lm, s_item;
    __shared__ int s_ew, s_df, s_bad;
    typedef unsigned long long u64;
    const int tid = threadIdx.x, nt = 1024;
    const int m = P.m, n = P.n, NP = P.NP;
    char* const lds = reinterpret_cast<char*>(osd_smem);
    unsigned* const act = reinterpret_cast<unsigned*>(lds + Wk.lds_act);      // [m]: the rows a block updates
    u64* const table = reinterpret_cast<u64*>(lds);                        // region 0: keys -> pos -> table
    u64* const Qs = reinterpret_cast<u64*>(lds + Wk.lds_region0);          // [8][wc_max]
    u64* keys;
    int *idx, *pos;
    uint8_t* sol;
    if (Wk.keys_in_lds) {
        keys = reinterpret_cast<u64*>(lds);
        pos = reinterpret_cast<int*>(lds);
        idx = reinterpret_cast<int*>(lds + Wk.lds_region0 + (size_t)8 * Wk.wc_max * 8);
        sol = reinterpret_cast<uint8_t*>(idx + NP);
    } else {
        keys = Wk.keys + (size_t)blockIdx.x * NP;
        pos = reinterpret_cast<int*>(keys);
        idx = Wk.idx + (size_t)blockIdx.x * NP;
        sol = Wk.sol + (size_t)blockIdx.x * n;
    }
    u64* const At = reinterpret_cast<u64*>(Wk.At) + (size_t)blockIdx.x * Wk.wc_max * m;
    OSD_T0();
    if (tid < 3) s_piv[tid] = ~0u;
    if (tid < 2) s_nact[tid] = 0u;
    unsigned col_ctr = 0;                     // columns swept so far: slot col_ctr % 3 of s_piv is the live one
    unsigned blk_ctr = 0;                     // blocks with pivots so far: s_nact[blk_ctr & 1] counts this one's rows
    __syncthreads();

    const long long total = P.count_ptr ? *P.count_ptr : P.count;
    // (syndromes one at a time from a counter: a sweep takes anything between 0.3 and 20 ms)
    for (;;) {
        if (tid == 0) s_item = atomicAdd(Wk.next, 1ull);
        __syncthreads();
        const long long item = (long long)s_item;
        if (item >= total) break;
        const long long rec = P.list ? P.list[item] : item;
        const double* llr = P.llr + rec * n;
        const uint8_t* hard = P.hard + rec * n;
        const uint8_t* syn = P.syndromes + rec * m;
        // ---- 1. ordering = argsort(|llr|), ties by column index                    OSD.py:10-11
        for (int i = tid; i < NP; i += nt) {
            keys[i] = i < n ? osd_order_key(llr[i]) : ~0ull;
            idx[i] = i;
        }
        for (int i = tid; i < n; i += nt) sol[i] = hard[i] & 1u;
        __syncthreads();
        for (int k = 2; k <= NP; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < NP / 2; t += nt) {
                    const int lo = ((t / j) * (2 * j)) + (t % j);
                    const int hi = lo + j;
                    const bool up = (lo & k) == 0;
                    const u64 ka = keys[lo], kb = keys[hi];
                    const int ia = idx[lo], ib = idx[hi];
                    if (osd_less(kb, ib, ka, ia) == up) {
                        keys[lo] = kb; keys[hi] = ka; idx[lo] = ib; idx[hi] = ia;
                    }
                }
                __syncthreads();
            }
        }
        OSD_T(0);
        int pc[RPT];                          // pivot column (original index) of this thread's rows, -1: none yet
        int Wc = 0;
        unsigned sb = 0;                      // bit i: reduced syndrome bit of row tid + i * nt
        for (int K = Wk.k_first < n ? Wk.k_first : n;; K = K < n / 4 ? 4 * K : n) {
            Wc = ((K + 63) >> 6) + 1;         // word planes: K sorted columns, then the syndrome bit
            // ---- 2. A = [H[:, order[:K]] | residual syndrome]                       OSD.py:7-11
            for (int k = tid; k < n; k += nt) pos[idx[k]] = k;
            __syncthreads();
            sb = 0;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int r = tid + i * nt;
                pc[i] = -1;
                if (r >= m) continue;
                for (int w = 0; w < Wc - 1; ++w) At[(size_t)w * m + r] = 0ull;
                unsigned par = syn[r] & 1u;
                for (int e = P.row_ptr[r]; e < P.row_ptr[r + 1]; ++e) {
                    const int c = P.col_idx[e];
                    par ^= sol[c];
                    const int k = pos[c];
                    if (k < K) At[(size_t)(k >> 6) * m + r] |= 1ull << (k & 63);
                }
                At[(size_t)(Wc - 1) * m + r] = par;
                sb |= par << i;
            }
            __syncthreads();                  // (pos is dead from here on: the table takes its place)
            OSD_T(1);
            // ---- 3. Gauss-Jordan over the sorted columns, a block of T at a time    OSD.py:31-72
            // The sweep ends at the rank of H (:42-43) -- or as soon as no row without a pivot has a 1 left in
            // the syndrome column: pivots found from there on would be chosen with a reduced syndrome bit of 0,
            // which XORs nothing into the syndrome bits above them and puts a 0 at their own column, so the
            // solution is already what the full sweep would leave.  (BP's residual syndromes are light: the
            // sweep typically ends after a small part of the columns.)
            int rank = 0, k0 = 0;
            bool open_rows = true;
            while (k0 < K && rank < P.rank) {
                {
                    unsigned mine = 0;
#pragma unroll
                    for (int i = 0; i < RPT; ++i) mine |= (pc[i] < 0 ? 1u : 0u) & (sb >> i);
                    open_rows = __syncthreads_or((int)mine) != 0;
                    if (!open_rows) break;
                }
                const int wk = k0 >> 6, sh = k0 & 63, nw = Wc - wk;
                int T = 8;
                while (T > 1 && ((k0 & (T - 1)) || ((size_t)nw << T) * 8 > (size_t)Wk.lds_table)) T >>= 1;
                const int Tc = K - k0 < T ? K - k0 : T;
                unsigned bd[RPT];             // bits 0-7: the row's bits in the block's columns, 8-15: its set D
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int r = tid + i * nt;
                    bd[i] = r < m ? (unsigned)(At[(size_t)wk * m + r] >> sh) & ((1u << Tc) - 1u) : 0u;
                }
                int prow[8];
                bool any = false;
                OSD_T(2);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    prow[j] = -1;
                    if (j >= Tc || rank >= P.rank) continue;          // (uniform)
                    // (this wavefront's first candidate through ballot + readlane, then one LDS atomic per
                    // wavefront: a thousand lanes on one LDS word took 8 000 cycles per column)
                    unsigned cand = ~0u;
#pragma unroll
                    for (int i = RPT - 1; i >= 0; --i) {
                        const u64 b = __ballot(pc[i] < 0 && ((bd[i] >> j) & 1u));
                        if (b) {
                            const int l = __builtin_ctzll(b);
                            const unsigned v = (unsigned)__builtin_amdgcn_readlane((int)bd[i], l);
                            cand = ((unsigned)((tid & ~63) + l + i * nt) << 16) | v;
                        }
                    }
                    unsigned* const slot = &s_piv[col_ctr % 3u];
                    if ((tid & 63) == 0 && cand != ~0u) atomicMin(slot, cand);
                    __syncthreads();
                    const unsigned key = *slot;                       // first unused row with a 1 (:46-50)
                    if (tid == 0) s_piv[(col_ctr + 2u) % 3u] = ~0u;   // (the slot of two columns ahead: idle now)
                    ++col_ctr;
                    if (key == ~0u) continue;                         // depends on earlier columns (:52-53)
                    ++rank;
                    any = true;
                    const int p = (int)(key >> 16);
                    // (the pivot's byte into the low bits, pivot j plus the pivot's own pending set into D)
                    const unsigned upd = (key & 0xffffu) | (0x100u << j);
                    prow[j] = p;
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const int r = tid + i * nt;
                        if (r == p) pc[i] = idx[k0 + j];
                        else if ((bd[i] >> j) & 1u) bd[i] ^= upd;                           // :63-68, on the bytes
                    }
                }
                k0 += T;
                OSD_T(3);
                if (!any) continue;
                unsigned* const nact = &s_nact[blk_ctr & 1u];
                {
                    // the rows this block changes, as (row, D) words in LDS: the update below is spread over all
                    // threads (a block touches some 40 of the rows of a sparse matrix)
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const bool a = (bd[i] >> 8) != 0u;
                        const u64 b = __ballot(a);
                        if (b) {
                            unsigned base = 0;
                            if ((tid & 63) == 0) base = atomicAdd(nact, (unsigned)__builtin_popcountll(b));
                            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                            if (a) act[base + (unsigned)__builtin_popcountll(b & ((1ull << (tid & 63)) - 1ull))] =
                                       ((unsigned)(tid + i * nt) << 8) | (bd[i] >> 8);
                        }
                    }
                    if (tid == 0) s_nact[(blk_ctr + 1u) & 1u] = 0u;       // (the next block's counter: idle now)
                }
                ++blk_ctr;
                // the block's pivot rows as they were when the block began -> LDS
                for (int it = tid; it < 8 * nw; it += nt) {
                    const int j = it / nw, w = it - j * nw;
                    int p = -1;
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (q == j) p = prow[q];
                    Qs[it] = p >= 0 ? At[(size_t)(wk + w) * m + p] : 0ull;
                }
                __syncthreads();
                OSD_T(4);
                // every XOR combination of them
                if (T == 8) {
                    // an item: word w, the eight entries x = 8 xh .. 8 xh + 7 (Gray order over the low three rows)
                    for (int it = tid; it < nw * 32; it += nt) {
                        const int xh = it / nw, w = it - xh * nw;
                        u64 v = 0ull;
#pragma unroll
                        for (int j = 3; j < 8; ++j) if ((xh >> (j - 3)) & 1) v ^= Qs[j * nw + w];
                        const u64 q0 = Qs[w], q1 = Qs[nw + w], q2 = Qs[2 * nw + w];
                        u64* const t = table + (size_t)(xh * 8) * nw + w;
                        t[0] = v;                     v ^= q0;
                        t[(size_t)1 * nw] = v;        v ^= q1;
                        t[(size_t)3 * nw] = v;        v ^= q0;
                        t[(size_t)2 * nw] = v;        v ^= q2;
                        t[(size_t)6 * nw] = v;        v ^= q0;
                        t[(size_t)7 * nw] = v;        v ^= q1;
                        t[(size_t)5 * nw] = v;        v ^= q0;
                        t[(size_t)4 * nw] = v;
                    }
                } else {
                    for (int it = tid; it < (nw << T); it += nt) {
                        const int x = it / nw, w = it - x * nw;
                        u64 v = 0ull;
#pragma unroll
                        for (int j = 0; j < 8; ++j) if ((x >> j) & 1) v ^= Qs[j * nw + w];
                        table[it] = v;
                    }
                }
                __syncthreads();
                OSD_T(5);
                {
#pragma unroll
                    for (int i = 0; i < RPT; ++i)
                        if (bd[i] >> 8) sb ^= ((unsigned)table[(size_t)(bd[i] >> 8) * nw + nw - 1] & 1u) << i;
                    const int na = (int)*nact, items = na * nw;
                    OSD_STAT(1, tid == 0 ? na : 0); OSD_STAT(2, tid == 0 ? items : 0);
                    for (int it0 = tid; it0 < items; it0 += 4 * nt) {
                        u64 v[4], t[4];
                        u64* a[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int it = it0 + q * nt;
                            if (it < items) {
                                const int w = it / na;
                                const unsigned e = act[it - w * na];
                                a[q] = At + (size_t)(wk + w) * m + (e >> 8);
                                t[q] = table[(size_t)(e & 0xffu) * nw + w];
                                v[q] = *a[q];
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (it0 + q * nt < items) *a[q] = v[q] ^ t[q];
                    }
                }
                // (the barrier at the top of the next block comes before anyone reads a row again or
                // overwrites Qs / the table)
                OSD_SYNC();
                if (tid == 0) { OSD_STAT(0, 1); }
                OSD_T(6);
            }
            OSD_STAT(7, tid == 0 ? 1 : 0);
#ifdef QBP_OSD_TIMING
            if (tid == 0) atomicMax(&g_osd_stat[3], (unsigned long long)k0);
#endif
            if (open_rows) {                  // (the loop ended on one of its other conditions: look once more)
                unsigned mine = 0;
#pragma unroll
                for (int i = 0; i < RPT; ++i) mine |= (pc[i] < 0 ? 1u : 0u) & (sb >> i);
                open_rows = __syncthreads_or((int)mine) != 0;
            }
            OSD_STAT(4, tid == 0 ? k0 : 0); OSD_STAT(5, tid == 0 ? rank : 0); OSD_STAT(6, tid == 0 && !open_rows ? 1 : 0);
            if (rank >= P.rank || !open_rows || K >= n) {
                if (open_rows && tid == 0) osd_flag_inconsistent(P, rec);
                break;
            }
            __syncthreads();                  // (next sweep: pos overwrites the table)
        }
        // ---- 4. e[pivot column] = reduced syndrome bit; solution = hard + e         OSD.py:14-26
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            // (sb: the thread's running copy of its rows' bits in the syndrome plane)
            if (pc[i] >= 0 && ((sb >> i) & 1u)) sol[pc[i]] ^= 1u;                     // distinct pivot columns
        }
        if (tid == 0) { s_lm = 0ull; s_ew = 0; s_df = 0; s_bad = 0; }
        __syncthreads();
        if (P.solution)
            for (int i = tid; i < n; i += nt) P.solution[rec * n + i] = sol[i];
        if (P.errors) {
            const uint8_t* err = P.errors + rec * n;
            u64 lm = 0ull;
            int ew = 0;
            unsigned df = 0, bad = 0;
            for (int i = tid; i < n; i += nt) {
                const unsigned e = err[i] & 1u;
                const unsigned res = sol[i] ^ e;
                ew += (int)e;
                df |= res;
                if (res) lm ^= P.lx_cols[i];
            }
            for (int r = tid; r < m; r += nt) {
                unsigned par = syn[r] & 1u;
                for (int e = P.row_ptr[r]; e < P.row_ptr[r + 1]; ++e) par ^= sol[P.col_idx[e]];
                bad |= par;
            }
            if (lm) atomicXor(&s_lm, lm);
            if (ew) atomicAdd(&s_ew, ew);
            if (df) atomicOr(&s_df, 1);
            if (bad) atomicOr(&s_bad, 1);
            __syncthreads();
            if (tid == 0) {
                auto add = [&](int i) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(P.counters + i), 1ull);
                };
                const bool logical = s_lm != 0ull;
                if (!s_bad && !logical && s_df) add(5);
                if (logical) {
                    add(1);
                    add(s_ew < P.half_distance ? 3 : 4);
                    add(8);
                }
                if (!s_df) add(9);
                if (s_bad) add(10);
            }
        }
        __syncthreads();
        OSD_T(7);
    }
    OSD_TEND();
}

#endif  // QBP_DEFINE_KERNELS

}  // namespace qbp
