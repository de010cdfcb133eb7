// OSD-0 post-processing on the GPU: decoding/OSD.py:3-72 (performOSD + gf2_elimination).
//
// One wavefront per syndrome.  The reference permutes the columns of H by ascending |LLR|, runs a
// Gauss-Jordan elimination with row swaps and reads the solution off the pivot columns.  The
// solution only depends on WHICH columns become pivots (the greedy, first-independent-columns
// basis in reliability order) -- not on which row serves as the pivot -- so this kernel never
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
        for (int r = lane; r < m; r += 64) {
            for (int w = 0; w < W; ++w) A[r * RS + w] = P.hbits[r * W + w];
            unsigned par = syn[r] & 1u;
            for (int e = P.row_ptr[r]; e < P.row_ptr[r + 1]; ++e) par ^= sol[P.col_idx[e]];
            A[r * RS + W] = par;
            pivcol[r] = -1;
        }
        __syncthreads();
        // ---- 3. Gauss-Jordan over the columns in reliability order            OSD.py:31-72
        // Lane l owns rows l, l + 64, ...: per column one LDS read per owned row decides both the
        // pivot search (ballot) and which rows take the XOR; the pivot row is read once (broadcast)
        // into registers when it fits (WW = words per row incl. the syndrome word, compile time).
        int rank = 0;
        unsigned used = 0;                           // bit i: row lane + 64 i already serves as a pivot row
        for (int k = 0; k < n && rank < P.rank; ++k) {   // :42-43 stops at m rows; rank(H) <= m
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
                    }
                }
            } else {
                for (int r = lane, i = 0; r < m; r += 64, ++i) {
                    if (r != p && ((has >> i) & 1u)) {
                        for (int w = 0; w <= W; ++w) A[r * RS + w] ^= A[p * RS + w];        // :63-68
                    }
                }
            }
            if (lane == 0) pivcol[p] = c;
            __syncthreads();
        }
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
// codes, where both kernels apply).
struct OsdBigWorkspace {
    uint32_t* At;                   // [grid][(W + 1) * m]
    int32_t* pivcol;                // [grid][m]
    uint8_t* sol;                   // [grid][n]
    unsigned long long* keys;       // [grid][NP]   (only when the keys do not fit LDS)
    int32_t* idx;                   // [grid][NP]
    int keys_in_lds;
};

#ifdef QBP_DEFINE_KERNELS   /* non-template kernel: defined in its translation unit only */
__global__ __launch_bounds__(256) void osd0_big_kernel(const OsdParams P, const OsdBigWorkspace Wk)
{
    extern __shared__ double osd_smem[];
    __shared__ int s_piv[2];        // (two slots, by column parity: see the pivot search)
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
        }
        __syncthreads();
        // ---- 3. Gauss-Jordan over the columns in reliability order                  OSD.py:31-72
        int rank = 0;
        for (int k = 0; k < n && rank < P.rank; ++k) {
            const int c = idx[k];
            const uint32_t* const colw = At + (size_t)(c >> 5) * m;
            const uint32_t bit = 1u << (c & 31);
            // (slot k & 1: a column without pivot leaves this iteration without a trailing barrier, so the
            // next column's reset must not touch the word the slower threads are still reading)
            int* const piv = &s_piv[k & 1];
            if (tid == 0) *piv = 0x7fffffff;
            __syncthreads();
            int mine = 0x7fffffff;
            for (int r = tid; r < m; r += nt)
                if ((colw[r] & bit) && pivcol[r] < 0) { mine = r; break; }     // this thread's first
            if (mine != 0x7fffffff) atomicMin(piv, mine);
            __syncthreads();
            const int p = *piv;                       // the first unused row with a 1 (:46-50)
            if (p == 0x7fffffff) continue;            // column depends on earlier ones (:52-53)
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

#endif  // QBP_DEFINE_KERNELS

}  // namespace qbp
