// libqbp.so -- host side of the C ABI declared in include/qbp.h (gfx950 only, no CPU fallback).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/qbp.h"
#include "qbp_kernels.hpp"
#include "qbp_osd.hpp"
#include "qbp_generic.hpp"
#include "qbp_stream.hpp"
#include "qbp_hist.hpp"
#include "qbp_launch.hpp"

static_assert(QBP_NUM_COUNTERS == qbp::NUM_COUNTERS, "counter layout");

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// No C++ exception crosses the C ABI: EVERY extern "C" entry point is a function-try-block ending in
// QBP_ABI_CATCH (std::bad_alloc -> QBP_E_NOMEM: tests/test_host_cpu.py forces one).
#define QBP_ABI_CATCH                                                                      \
    catch (const std::bad_alloc&) { return fail(QBP_E_NOMEM, "out of host memory"); }     \
    catch (const std::exception& ex) { return fail(QBP_E_INVALID, "internal error: %s", ex.what()); } \
    catch (...) { return fail(QBP_E_INVALID, "internal error"); }

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(QBP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                               \
    } while (0)

using qbp::DC_SMALL; using qbp::DV_SMALL; using qbp::DC_WIDE; using qbp::DV_WIDE;

// Makes `device` current for the lifetime of the object and restores the caller's device
// afterwards (entry points must not leave a host thread on another GPU than they found it on).
struct DeviceScope {
    int prev = -1;
    hipError_t err;
    explicit DeviceScope(int device)
    {
        err = hipGetDevice(&prev);
        if (err != hipSuccess) { prev = -1; err = hipSuccess; }
        if (prev != device) err = hipSetDevice(device); else prev = -1;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t count)
    {
        if (count <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) cap = count;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct qbp_handle {
    int device = 0;
    int m = 0, n = 0, E = 0;
    int max_row_deg = 0, max_col_deg = 0;
    int num_cu = 0;
    bool fused_ok = false;
    int dc = DC_SMALL, dv = DV_SMALL;   // instantiation used by this matrix
    bool padded = false;
    std::vector<int32_t> row_ptr, col_idx;
    // device tables for the fused kernel
    DevBuf<int32_t> d_tab_var;
    DevBuf<uint16_t> d_tab_nbr;
    DevBuf<uint32_t> d_tab_writer;
    DevBuf<int32_t> d_iso;
    int n_iso = 0;
    DevBuf<unsigned long long> d_work_counter;
    // options
    int opt_slots = 0, opt_blocks_per_cu = 0;
    // last launch configuration (introspection)
    int last_threads = 0, last_lds = 0, last_grid = 0, last_kernel = 0;
    // scratch for the host-pointer entry points
    hipStream_t stream = nullptr;
    DevBuf<uint8_t> d_syn, d_hard, d_conv;
    DevBuf<int32_t> d_iters;
    DevBuf<double> d_llr, d_prior, d_mathx, d_mathy, d_part, d_edges;
    DevBuf<unsigned long long> d_hist;
    DevBuf<unsigned long long> d_lx_cols;
    DevBuf<long long> d_counters;
    std::vector<uint8_t> lx_cache;   // last uploaded Lx (host copy) to skip re-uploads
    int lx_cache_k = -1;
    // general-H kernel
    DevBuf<int32_t> d_col_ptr, d_col_edge;
    DevBuf<double> d_wsQ, d_wsR, d_wsV;
    DevBuf<uint8_t> d_wsC;
    int opt_force_generic = 0;
    int opt_forced_two_barriers = 0;   // 1: forced-iteration launches keep the second barrier of the iteration (A/B, tests)
    int last_one_barrier = 0;
    int opt_early_exit_full_wg = 0;    // 1: early-exit launches use 16-wavefront workgroups like forced ones (A/B)
    int opt_no_r0_table = 0;           // 1: early-exit launches compute the first check step like every other (A/B, tests)
    int opt_kernel = 0;             // 0 auto, 1 on-chip, 2 general-H (workgroup per syndrome), 3 streaming
    DevBuf<uint8_t> d_wsS;           // streaming kernel: transposed syndromes; general-H Monte-Carlo: syndromes
    DevBuf<uint8_t> d_wsE;           // general-H Monte-Carlo: sampled errors
    DevBuf<int32_t> d_srow, d_srow_e0, d_srow_deg, d_svar, d_sedge;   // weight-class tables
    DevBuf<int32_t> d_epos, d_cpos, d_long_edge_row;
    DevBuf<int32_t> d_vpos, d_vrow, d_lcol_ptr;      // general-H kernel: column-class tables
    DevBuf<double> d_wsL, d_prior_sorted;
    int gcol_base[qbp::GENERIC_MAX_COL_CLASS + 2] = {0};
    int rpad_off[qbp::GENERIC_MAX_ROW_CLASS + 2] = {0}, cpad_off[qbp::GENERIC_MAX_COL_CLASS + 2] = {0};
    int opt_threads = 0;            // general-H kernel: threads per workgroup (0 = auto)
    int opt_no_lds_tables = 0;      // general-H kernel: keep the variable step's tables in L2 (A/B)
    int opt_no_r_split = 0;         // general-H kernel: no part of R in LDS when the messages do not fit (A/B)
    int opt_mem = 0;                // general-H kernel: 0 auto, 1 messages in global memory, 2 half of R in LDS (tests)
    int row_base[qbp::GENERIC_MAX_ROW_CLASS + 2] = {0};
    int row_off[qbp::STREAM_MAX_ROW_CLASS + 3] = {0};
    int col_off[qbp::STREAM_MAX_COL_CLASS + 3] = {0}, col_edge_base[qbp::STREAM_MAX_COL_CLASS + 2] = {0};
    // pinned, device-mapped staging for small host-pointer calls (zero-copy: no hipMemcpy at all)
    void* pin_host = nullptr;
    void* pin_dev = nullptr;
    size_t pin_bytes = 0;
    // two pinned staging buffers for the outputs of large host-pointer calls (device -> pinned by DMA
    // at PCIe rate, pinned -> caller's pageable array by memcpy, one chunk behind)
    void* stage[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;
    // OSD-0
    bool osd_ok = false;            // fits the one-wavefront kernel (matrix rows in LDS)
    int opt_osd_big = 0;            // QBP_OPT_OSD_BIG
    bool osd_big_ready = false;     // tables of the workgroup-per-syndrome kernel built (lazily)
    int osd_W = 0, osd_NP = 0, osd_lds = 0, osd_rank = 0;
    DevBuf<uint32_t> d_osd_At;
    DevBuf<int32_t> d_osd_piv, d_osd_idx, d_osd_posn;
    DevBuf<unsigned long long> d_osd_next;   // work counter of osd0_blocked_kernel
    DevBuf<long long> d_osd_redo;   // [0]: count, [1..]: records whose OSD sweep found the syndrome inconsistent
    DevBuf<uint8_t> d_osd_sol;
    DevBuf<unsigned long long> d_osd_keys;
    DevBuf<uint32_t> d_hbits;
    DevBuf<int32_t> d_row_ptr, d_col_idx;
    DevBuf<uint8_t> d_sol;
    // the order-dependent tables once more, for column sums in the order of a Fortran-ordered dense R
    // (QBP_FLAG_DENSE_F_COLSUM; built at the first such call)
    struct ColumnOrderTables {
        int state = 0;               // 0 not built, 1 ready, -1 not expressible (QBP_E_UNSUPPORTED)
        DevBuf<uint16_t> tab_nbr;
        DevBuf<uint32_t> tab_writer;
        DevBuf<int32_t> col_edge, sedge, vpos, vrow;
    } f_order;
    // Monte-Carlo + OSD failure records
    DevBuf<long long> d_fail_list;
    DevBuf<unsigned long long> d_fail_count;
    DevBuf<uint8_t> d_fail_syn, d_fail_hard, d_fail_err;
    DevBuf<double> d_fail_llr;
};

namespace {

using qbp::FusedParams;

using qbp::LaunchCfg;

// Dynamic LDS of one workgroup of the fused kernel (the carve is documented in qbp_kernels.hpp)
size_t fused_lds_bytes(int dc, int m, int n, int S, bool two_copies = false, bool r0_table = false)
{
    const size_t slot_stride = (size_t)dc * m + 2;
    size_t lds = (size_t)qbp::NP_LDS_BYTES +      // tables of tanh / arctanh (qbp_math.hpp), at the start
                 (two_copies ? (size_t)qbp::FUSED_R2_OFF_BYTES : 0) + (r0_table ? (size_t)2 * dc * m * 8 : 0) +
                 ((size_t)S * slot_stride + (size_t)dc * m + 3 * (size_t)S) * 8 +
                 (6 * (size_t)S + 1 + (size_t)S * qbp::NUM_COUNTERS + (size_t)dc * m) * 4 +
                 2 * (size_t)S * (((size_t)n + 3) / 4) * 4;     // err_lds[2][S][n4] (Monte-Carlo builds)
    return (lds + 15) & ~(size_t)15;
}

// Everything qbp_create derives from the CSR arrays on the host (no device involved; qbp_plan
// exposes it so that the CPU test-suite can check the tables).
struct HostTables {
    int max_row = 0, max_col = 0, dc = DC_SMALL, dv = DV_SMALL;
    bool fused_ok = false, padded = false;
    std::vector<int32_t> tab_var;      // [dc][m]
    std::vector<uint16_t> tab_nbr;     // [dc][dv][m]
    std::vector<uint32_t> tab_writer;  // [m]
    std::vector<int32_t> iso, col_ptr, col_edge;
    // streaming kernel: checks / variables sorted by weight class (qbp_stream.hpp)
    std::vector<int32_t> srow, srow_e0, srow_deg, svar, sedge;
    // general-H kernel: class-blocked transposed message layout (qbp_generic.hpp)
    std::vector<int32_t> epos, cpos, long_edge_row;
    std::vector<int32_t> vpos, vrow, lcol_ptr;         // general-H kernel, variables by column class
    int gcol_base[qbp::GENERIC_MAX_COL_CLASS + 2] = {0};
    int row_base[qbp::GENERIC_MAX_ROW_CLASS + 2] = {0};
    int row_off[qbp::STREAM_MAX_ROW_CLASS + 3] = {0};
    int col_off[qbp::STREAM_MAX_COL_CLASS + 3] = {0}, col_edge_base[qbp::STREAM_MAX_COL_CLASS + 2] = {0};
};

// Order in which np.sum(R, axis=0) adds up the non-zero entries of one column of a FORTRAN-ordered dense
// (m, n) array R -- which is what the reference's dense single-syndrome forms reduce whenever the caller's
// H is Fortran-ordered, as the Hx of its code files is (every (m, n) temporary inherits the layout of
// `mask = H != 0`).  The column is contiguous then and numpy runs pairwise_sum over all m entries, zeros
// included (numpy/_core/src/umath/loops_utils.h.src): fewer than 8 rows left to right; up to 128 rows eight
// running sums by row mod 8, combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the m mod 8 trailing rows;
// more rows split in halves (the first a multiple of 8) recursively.  x + 0.0 is exact, so only the
// association of the non-zero entries matters: simulated here symbolically.  Returns false when that
// association is not a left-to-right sum of SOME ordering of the entries (possible from 4 entries on:
// (a + b) + (c + d)); `order` = that ordering as indices into the column's ascending-check list.
struct FSumNode { int leaf, l, r; };
static int fsum_add(std::vector<FSumNode>& pool, int a, int b)
{
    if (a < 0) return b;
    if (b < 0) return a;
    pool.push_back({-1, a, b});
    return (int)pool.size() - 1;
}
static int fsum_range(std::vector<FSumNode>& pool, const std::vector<int>& rows, int lo, int hi)
{
    const int n = hi - lo;
    int a = 0;
    while (a < (int)rows.size() && rows[a] < lo) ++a;
    int b = a;
    while (b < (int)rows.size() && rows[b] < hi) ++b;
    if (a == b) return -1;
    if (n < 8) {
        int res = -1;
        for (int i = a; i < b; ++i) res = fsum_add(pool, res, i);
        return res;
    }
    if (n <= 128) {
        int r[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        const int body = n - n % 8;
        int i = a;
        for (; i < b && rows[i] - lo < body; ++i) r[(rows[i] - lo) % 8] = fsum_add(pool, r[(rows[i] - lo) % 8], i);
        int res = fsum_add(pool, fsum_add(pool, fsum_add(pool, r[0], r[1]), fsum_add(pool, r[2], r[3])),
                           fsum_add(pool, fsum_add(pool, r[4], r[5]), fsum_add(pool, r[6], r[7])));
        for (; i < b; ++i) res = fsum_add(pool, res, i);
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return fsum_add(pool, fsum_range(pool, rows, lo, lo + n2), fsum_range(pool, rows, lo + n2, hi));
}
static bool fsum_flatten(const std::vector<FSumNode>& pool, int node, std::vector<int>& out)
{
    const FSumNode& x = pool[node];
    if (x.leaf >= 0) { out.push_back(x.leaf); return true; }
    const bool ll = pool[x.l].leaf >= 0, rl = pool[x.r].leaf >= 0;
    if (rl) { if (!fsum_flatten(pool, x.l, out)) return false; out.push_back(pool[x.r].leaf); return true; }
    if (ll) { if (!fsum_flatten(pool, x.r, out)) return false; out.push_back(pool[x.l].leaf); return true; }
    return false;
}
static bool dense_f_column_order(const std::vector<int>& rows, int m, std::vector<int>& order)
{
    std::vector<FSumNode> pool;
    for (int i = 0; i < (int)rows.size(); ++i) pool.push_back({i, -1, -1});
    order.clear();
    if (rows.empty()) return true;
    const int root = fsum_range(pool, rows, 0, m);
    return fsum_flatten(pool, root, order) && order.size() == rows.size();
}

// col_order: 0 = every column's entries in ascending check order (np.sum(R, axis=0) of a C-ordered R adds row
// by row); 1 = the order of a Fortran-ordered R (dense_f_column_order).  All kernels add a column's messages
// left to right in the order of these tables.
int build_tables(const int32_t* row_ptr, const int32_t* col_idx, int m, int n, HostTables& T, int col_order = 0)
{
    if (!row_ptr || m < 0 || n < 0) return fail(QBP_E_INVALID, "bad matrix arguments");
    if (m == 0 || n == 0) return fail(QBP_E_INVALID, "empty matrix (%d x %d)", m, n);
    if (row_ptr[0] != 0) return fail(QBP_E_INVALID, "row_ptr[0] must be 0");
    const int E = row_ptr[m];
    if (E < 0 || (E > 0 && !col_idx)) return fail(QBP_E_INVALID, "bad CSR arrays");
    // column lists in ascending check order: (check, position in that check's row)
    std::vector<std::vector<std::pair<int, int>>> cols(n);
    for (int c = 0; c < m; ++c) {
        if (row_ptr[c + 1] < row_ptr[c]) return fail(QBP_E_INVALID, "row_ptr not monotone at %d", c);
        T.max_row = std::max(T.max_row, row_ptr[c + 1] - row_ptr[c]);
        for (int e = row_ptr[c]; e < row_ptr[c + 1]; ++e) {
            if (col_idx[e] < 0 || col_idx[e] >= n)
                return fail(QBP_E_INVALID, "column index %d out of range in row %d", col_idx[e], c);
            if (e > row_ptr[c] && col_idx[e] <= col_idx[e - 1])
                return fail(QBP_E_INVALID, "columns of row %d not strictly ascending", c);
            cols[col_idx[e]].push_back({c, e - row_ptr[c]});
        }
    }
    for (int v = 0; v < n; ++v) {
        T.max_col = std::max(T.max_col, (int)cols[v].size());
        if (cols[v].empty()) T.iso.push_back(v);
    }
    if (col_order == 1) {
        std::vector<int> rows, order;
        for (int v = 0; v < n; ++v) {
            rows.clear();
            for (const auto& ce : cols[v]) rows.push_back(ce.first);
            if (!dense_f_column_order(rows, m, order))
                return fail(QBP_E_UNSUPPORTED,
                            "column %d (%zu entries): numpy's pairwise sum over a Fortran-ordered dense column "
                            "associates them as a balanced tree, which the kernels' left-to-right sums cannot "
                            "express", v, rows.size());
            std::vector<std::pair<int, int>> perm;
            for (int i : order) perm.push_back(cols[v][i]);
            cols[v] = perm;
        }
    }
    if (T.max_row <= DC_SMALL && T.max_col <= DV_SMALL) { T.dc = DC_SMALL; T.dv = DV_SMALL; }
    else { T.dc = DC_WIDE; T.dv = DV_WIDE; }
    T.fused_ok = (m <= qbp::FUSED_MAX_THREADS) && T.max_row <= T.dc && T.max_col <= T.dv &&
                 fused_lds_bytes(T.dc, m, n, 1) <= 160 * 1024;
    if (T.fused_ok) {
        const int DC = T.dc, DV = T.dv, zoff = DC * m;
        T.tab_var.assign((size_t)DC * m, -1);
        T.tab_nbr.assign((size_t)DC * DV * m, (uint16_t)zoff);
        T.tab_writer.assign(m, 0u);
        for (int c = 0; c < m; ++c) {
            const int deg = row_ptr[c + 1] - row_ptr[c];
            if (deg < DC) T.padded = true;
            for (int j = 0; j < deg; ++j) {
                const int v = col_idx[row_ptr[c] + j];
                T.tab_var[(size_t)j * m + c] = v;
                const auto& col = cols[v];
                for (size_t k = 0; k < col.size(); ++k)
                    T.tab_nbr[((size_t)j * DV + k) * m + c] = (uint16_t)(col[k].second * m + col[k].first);
                if (col[0].first == c) T.tab_writer[c] |= 1u << j;     // (one lane per variable writes its outputs)
            }
        }
    }
    // CSC (edge ids per column, ascending check) for the general-H kernel
    T.col_ptr.assign((size_t)n + 1, 0);
    T.col_edge.assign((size_t)std::max(E, 1), 0);
    for (int v = 0; v < n; ++v) T.col_ptr[v + 1] = T.col_ptr[v] + (int)cols[v].size();
    for (int v = 0; v < n; ++v)
        for (size_t k = 0; k < cols[v].size(); ++k)
            T.col_edge[T.col_ptr[v] + k] = row_ptr[cols[v][k].first] + cols[v][k].second;
    // weight classes of the streaming kernel: rows 0 .. 8 and > 8, columns 0 .. 4 and > 4
    constexpr int RC = qbp::STREAM_MAX_ROW_CLASS, CC = qbp::STREAM_MAX_COL_CLASS;
    for (int k = 0; k <= RC + 1; ++k) {
        T.row_off[k] = (int)T.srow.size();
        for (int c = 0; c < m; ++c) {
            const int deg = row_ptr[c + 1] - row_ptr[c];
            if (std::min(deg, RC + 1) != k) continue;
            T.srow.push_back(c);
            T.srow_e0.push_back(row_ptr[c]);
            T.srow_deg.push_back(deg);
        }
    }
    T.row_off[RC + 2] = m;
    static_assert(qbp::GENERIC_MAX_ROW_CLASS == qbp::STREAM_MAX_ROW_CLASS, "shared row classes");
    T.epos.assign((size_t)std::max(E, 1), 0);
    T.cpos.assign((size_t)std::max(E, 1), 0);
    {
        int base = 0;
        for (int k = 1; k <= RC; ++k) {                 // block of the weight-k checks: [entry j][check i]
            const int cnt = T.row_off[k + 1] - T.row_off[k];
            T.row_base[k] = base;
            for (int i = 0; i < cnt; ++i)
                for (int j = 0; j < k; ++j) T.epos[T.srow_e0[T.row_off[k] + i] + j] = base + j * cnt + i;
            base += k * cnt;
        }
        T.row_base[RC + 1] = base;                      // longer checks: entries contiguous
        for (int i = T.row_off[RC + 1]; i < T.row_off[RC + 2]; ++i)
            for (int j = 0; j < T.srow_deg[i]; ++j) {
                T.epos[T.srow_e0[i] + j] = base++;
                T.long_edge_row.push_back(i - T.row_off[RC + 1]);
            }
    }
    for (int q = 0; q < E; ++q) T.cpos[q] = T.epos[T.col_edge[q]];
    for (int k = 0; k <= CC + 1; ++k) {
        T.col_off[k] = (int)T.svar.size();
        T.col_edge_base[k] = (int)T.sedge.size();
        for (int v = 0; v < n; ++v) {
            if (std::min<int>((int)cols[v].size(), CC + 1) != k) continue;
            T.svar.push_back(v);
            if (k >= 1 && k <= CC)
                for (int q = T.col_ptr[v]; q < T.col_ptr[v + 1]; ++q) T.sedge.push_back(T.col_edge[q]);
        }
    }
    T.col_off[CC + 2] = n;
    if (T.sedge.empty()) T.sedge.push_back(0);
    // general-H kernel: per column entry the message position and the SORTED position of its check,
    // blocked and transposed by column-weight class (qbp_generic.hpp); long columns contiguous
    static_assert(qbp::GENERIC_MAX_COL_CLASS == qbp::STREAM_MAX_COL_CLASS, "shared column classes");
    {
        std::vector<int32_t> wpos((size_t)m, 0);
        for (int w = 0; w < m; ++w) wpos[T.srow[w]] = w;
        T.vpos.assign((size_t)std::max(E, 1), 0);
        T.vrow.assign((size_t)std::max(E, 1), 0);
        int base = 0;
        for (int k = 1; k <= CC; ++k) {
            const int cnt = T.col_off[k + 1] - T.col_off[k];
            T.gcol_base[k] = base;
            for (int i = 0; i < cnt; ++i) {
                const int v = T.svar[T.col_off[k] + i];
                for (int j = 0; j < k; ++j) {
                    T.vpos[(size_t)base + (size_t)j * cnt + i] = T.cpos[T.col_ptr[v] + j];
                    T.vrow[(size_t)base + (size_t)j * cnt + i] = wpos[cols[v][j].first];
                }
            }
            base += k * cnt;
        }
        T.gcol_base[CC + 1] = base;
        for (int i = T.col_off[CC + 1]; i < T.col_off[CC + 2]; ++i) {
            const int v = T.svar[i];
            T.lcol_ptr.push_back(base);
            for (size_t j = 0; j < cols[v].size(); ++j) {
                T.vpos[base] = T.cpos[T.col_ptr[v] + (int)j];
                T.vrow[base] = wpos[cols[v][j].first];
                ++base;
            }
        }
        T.lcol_ptr.push_back(base);
    }
    return QBP_OK;
}

int make_cfg(qbp_handle* h, long long B, LaunchCfg* cfg, bool forced = false, bool mc = false)
{
    const int m = h->m;
    int S = h->opt_slots;
    if (S <= 0) {
        // 16 wavefronts per workgroup (4 per SIMD at the kernel's 128-VGPR budget) was the fastest
        // geometry measured (profiles/r01_tune.txt); small batches spread over the CUs instead.
        S = std::max(1, qbp::FUSED_MAX_THREADS / std::max(m, 1));
        // With early exit the slots of a workgroup are in different iterations, yet every one of them
        // waits at the workgroup's barriers for the slowest (a slot in its first iteration copies six
        // table entries, its neighbour evaluates six tanh / division / atanh chains): two workgroups
        // of 8 wavefronts per CU -- half the slots behind each barrier, and one workgroup computing
        // while the other synchronises or writes results -- are 2 - 18 % faster on every early-exit
        // workload measured, although [[288,12,18]] then runs 6 syndromes per CU instead of 7
        // (profiles/r02_ab_work_chunk.txt, second part).  Not for the (8, 4) shape (m > 512).
        if (!forced && !h->opt_early_exit_full_wg && (qbp::FUSED_MAX_THREADS / 2) / std::max(m, 1) >= 1 && S >= 2)
            S = (qbp::FUSED_MAX_THREADS / 2) / m;
        const long long spread = (B + h->num_cu - 1) / std::max(h->num_cu, 1);
        if (spread < S) S = (int)std::max<long long>(spread, 1);
    }
    S = std::min(S, std::max(1, qbp::FUSED_MAX_THREADS / std::max(m, 1)));
    if ((long long)S > B) S = (int)std::max<long long>(B, 1);
    cfg->dc = h->dc;
    cfg->slot_stride = h->dc * m + 2;
    while (S > 1 && fused_lds_bytes(h->dc, m, h->n, S) > 160 * 1024) --S;   // LDS-limited shapes
    // forced-iteration decode: the one-barrier kernel when two copies of the messages fit (the first one
    // below the constant offset of the second) without giving up a slot
    const bool two = forced && !mc && !h->opt_forced_two_barriers && h->dc == DC_SMALL &&
                     (size_t)S * cfg->slot_stride * 8 <= (size_t)qbp::FUSED_R2_OFF_BYTES &&
                     fused_lds_bytes(h->dc, m, h->n, S, true) <= 160 * 1024;
    cfg->one_barrier = two ? 1 : 0;
    h->last_one_barrier = cfg->one_barrier;
    cfg->S = S;
    cfg->threads = std::min(qbp::FUSED_MAX_THREADS, ((S * m + 63) / 64) * 64);
    // early exit: the first check step's messages as an LDS table, when it fits beside S slots
    const bool r0 = !forced && !h->opt_no_r0_table && fused_lds_bytes(h->dc, m, h->n, S, false, true) <= 160 * 1024;
    cfg->r0_table = r0 ? 1 : 0;
    const size_t lds = fused_lds_bytes(h->dc, m, h->n, S, two, r0);
    if (lds > 160 * 1024) return fail(QBP_E_UNSUPPORTED, "LDS need %zu B exceeds 160 KiB", lds);
    cfg->lds_bytes = (int)lds;
    int per_cu = h->opt_blocks_per_cu;
    if (per_cu <= 0) {
        // resident workgroups per CU: 16 wavefronts fit at the 128-VGPR budget; also LDS.  At least
        // two workgroups per CU are queued: with early exit the second round evens out the tail
        // (measured +10 % at p = 0.01, neutral when every syndrome runs max_iter iterations).
        const int waves = cfg->threads / 64;
        const int resident = std::max(1, std::min((qbp::FUSED_MAX_THREADS / 64) / std::max(waves, 1), (int)(160 * 1024 / lds)));
        per_cu = 2 * resident;
    }
    const long long want = (B + S - 1) / S;
    cfg->grid = (int)std::max<long long>(1, std::min<long long>(want, (long long)h->num_cu * per_cu));
    h->last_threads = cfg->threads;
    h->last_lds = cfg->lds_bytes;
    h->last_grid = cfg->grid;
    return QBP_OK;
}

int check_decode_args(qbp_handle* h, long long B, int max_iter, int variant, bool need_fused = false)
{
    if (!h) return fail(QBP_E_INVALID, "null handle");
    if (B < 0) return fail(QBP_E_INVALID, "B must be >= 0 (got %lld)", B);
    if (max_iter < 1)
        return fail(QBP_E_INVALID, "max_iter must be >= 1 (got %d); the reference raises "
                                   "UnboundLocalError for maxIter=0", max_iter);
    if (variant < 0 || variant > 2) return fail(QBP_E_INVALID, "unknown variant %d", variant);
    if (need_fused && !h->fused_ok)
        return fail(QBP_E_UNSUPPORTED,
                    "H (m=%d, max row degree %d, max column degree %d) does not fit the on-chip "
                    "kernel (m <= 1024, row degree <= %d, column degree <= %d)",
                    h->m, h->max_row_deg, h->max_col_deg, DC_WIDE, DV_WIDE);
    return QBP_OK;
}

void fill_static(qbp_handle* h, FusedParams& P, const LaunchCfg& cfg, bool f_order = false)
{
    P.m = h->m; P.n = h->n;
    P.S = cfg.S; P.slot_stride = cfg.slot_stride;
    P.r0_table = cfg.r0_table;
    P.padded = h->padded ? 1 : 0;
    P.n_words4 = (h->n + 3) / 4;
    P.tab_var = h->d_tab_var.p;
    P.tab_nbr = f_order ? h->f_order.tab_nbr.p : h->d_tab_nbr.p;
    P.tab_writer = f_order ? h->f_order.tab_writer.p : h->d_tab_writer.p;
    P.iso_vars = h->d_iso.p; P.n_iso = h->n_iso;
    P.work_counter = h->d_work_counter.p;
}

// Column-sum order of a decode call (include/qbp.h, QBP_FLAG_DENSE_F_COLSUM*): clears the two flag bits and
// sets *f_order when the launch has to use the Fortran-order tables (built and uploaded at the first such call).
int resolve_column_order(qbp_handle* h, unsigned& flags, const double* host_prior, int* mode)
{
    *mode = 0;                               // 0 row by row, 1 Fortran order, 2 Fortran order at iteration 0 only
    const unsigned want = flags & (QBP_FLAG_DENSE_F_COLSUM | QBP_FLAG_DENSE_F_COLSUM_ITER0);
    flags &= ~(QBP_FLAG_DENSE_F_COLSUM | QBP_FLAG_DENSE_F_COLSUM_ITER0);
    if (!want) return QBP_OK;
    if (want == (QBP_FLAG_DENSE_F_COLSUM | QBP_FLAG_DENSE_F_COLSUM_ITER0) || (flags & QBP_FLAG_PAIRWISE_COLSUM))
        return fail(QBP_E_INVALID, "at most one column-sum order flag per call");
    if (want == QBP_FLAG_DENSE_F_COLSUM_ITER0 && host_prior) {
        // Iteration 0 does not depend on the order when all of a column's messages have the same magnitude r
        // -- equal priors on every edge and checks of equal weight -- and a column has at most three of them:
        // every partial sum is k r with |k| <= 3, and k r is exact for |k| <= 2 and rounded once, from the
        // exact value, for |k| = 3.  That is the reference's own use (rework/Alvarado.py, rework/main.py:
        // uniform priors on the code files' Hx) and needs no second table set in the launch.
        bool same = h->max_col_deg <= 3;
        const int d0 = h->m > 0 ? h->row_ptr[1] - h->row_ptr[0] : 0;
        for (int c = 0; same && c < h->m; ++c) same = (h->row_ptr[c + 1] - h->row_ptr[c]) == d0;
        if (same && h->E > 0) {
            const double p0 = host_prior[h->col_idx[0]];
            for (int e = 0; same && e < h->E; ++e) same = host_prior[h->col_idx[e]] == p0;
        }
        if (same) return QBP_OK;             // = row-by-row tables
    }
    auto& F = h->f_order;
    if (F.state == 0) {
        HostTables T;
        const int rc = build_tables(h->row_ptr.data(), h->col_idx.data(), h->m, h->n, T, 1);
        if (rc == QBP_E_UNSUPPORTED) { F.state = -1; return rc; }
        if (rc) return rc;
        hipError_t e1 = hipSuccess;
        auto up = [&](auto& buf, const auto& vec) {
            if (e1 != hipSuccess) return;
            e1 = buf.reserve(vec.size());
            if (e1 == hipSuccess && !vec.empty())
                e1 = hipMemcpy(buf.p, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice);
        };
        if (T.fused_ok) { up(F.tab_nbr, T.tab_nbr); up(F.tab_writer, T.tab_writer); }
        up(F.col_edge, T.col_edge); up(F.sedge, T.sedge); up(F.vpos, T.vpos); up(F.vrow, T.vrow);
        if (e1 != hipSuccess) return fail(QBP_E_HIP, "upload of the column-order tables failed: %s", hipGetErrorString(e1));
        F.state = 1;
    }
    if (F.state < 0)
        return fail(QBP_E_UNSUPPORTED, "the Fortran-order column sums of this matrix are not a left-to-right sum");
    *mode = want == QBP_FLAG_DENSE_F_COLSUM ? 1 : 2;
    return QBP_OK;
}

}  // namespace

// Launch geometry of the general-H kernel (one workgroup per syndrome at a time):
//   messages in LDS when they fit (16 E bytes + bookkeeping <= 160 KiB), else in a global workspace;
//   threads per workgroup: the checks of weight <= 8 in as few, as full passes as possible
//   (864 checks: 896 threads, one pass; 2592: 896 threads, three passes), at most 1024;
//   workgroups per CU: as many as fit 16 wavefronts (the kernel's 128-register budget) and the LDS.
struct GenericGeom { bool lds_msgs, lds_tables; int mem, r_split, threads, per_cu, grid; size_t lds; };

// inplace: sum-product without damping keeps one message array (qbp_generic.hpp): 8 E bytes instead of 16 E
static GenericGeom generic_geometry(const qbp_handle* h, long long B, bool inplace)
{
    GenericGeom g{};
    const int E1 = std::max(h->E, 1);
    constexpr size_t LDS_MAX = (size_t)160 * 1024;
    g.lds_msgs = qbp::generic_lds_bytes(h->m, E1, h->n, true, false, 0, inplace) <= LDS_MAX && h->opt_mem == 0;
    // the variable step's tables (prior, message positions) in LDS too when one workgroup per CU is
    // the geometry anyway and they fit beside (or instead of) the messages
    g.lds_tables = false;
    g.lds = qbp::generic_lds_bytes(h->m, E1, h->n, g.lds_msgs, false, 0, inplace);
    const int short_rows = std::max(1, h->row_off[qbp::GENERIC_MAX_ROW_CLASS + 1] - h->row_off[1]);
    const int work = std::max(h->rpad_off[qbp::GENERIC_MAX_ROW_CLASS + 1], 64);    // padded work items
    const int passes = (work + 1023) / 1024;
    int threads = (((work + passes - 1) / passes) + 63) / 64 * 64;
    int per_cu = std::max(1, 1024 / threads);
    if (g.lds) per_cu = (int)std::min<size_t>((size_t)per_cu, std::max<size_t>(1, ((size_t)160 * 1024) / g.lds));
    per_cu = std::min(per_cu, 8);
    // batches that cannot fill the chip: one wide workgroup per syndrome (latency of a single decode)
    if (B < (long long)h->num_cu * per_cu) {
        per_cu = 1;
        const int wide = std::max(std::max(short_rows, h->n / 2), 64);
        threads = std::min(1024, (wide + 63) / 64 * 64);
    }
    // messages in global memory: the variable step is bound by L2 / Infinity-Cache traffic, and more
    // threads per syndrome is what puts more of it in flight (2592 x 7776: 1.89e5 syndromes/s with 1024
    // threads against 1.72e5 with 896, the check step's exact fit)
    if (!g.lds_msgs && B >= (long long)h->num_cu) threads = 1024;
    if (h->opt_threads > 0) threads = std::min(1024, (h->opt_threads + 63) / 64 * 64);
    if (h->opt_blocks_per_cu > 0) per_cu = h->opt_blocks_per_cu;
    g.mem = g.lds_msgs ? qbp::GENERIC_MEM_LDS : qbp::GENERIC_MEM_GLOBAL;
    g.r_split = 0;
    if (h->opt_mem == 2) per_cu = 1;
    if (!g.lds_msgs && per_cu == 1 && !h->opt_no_r_split && h->opt_mem != 1) {
        // the messages do not fit: as much of R as the LDS holds (the rest, and Q, in the workspace)
        const size_t room = LDS_MAX - qbp::generic_lds_bytes(h->m, E1, h->n, false, false);
        int K = (int)std::min<size_t>((size_t)E1, room / 8);
        if (h->opt_mem == 2) K = std::min(K, std::max(1, E1 / 2));     // (tests: both sides of the split in use)
        if (K >= E1 / 4) { g.mem = qbp::GENERIC_MEM_SPLIT; g.r_split = K; }
        g.lds = qbp::generic_lds_bytes(h->m, E1, h->n, false, false, g.r_split);
    }
    if (per_cu == 1 && !h->opt_no_lds_tables && g.mem != qbp::GENERIC_MEM_SPLIT &&
        qbp::generic_lds_bytes(h->m, E1, h->n, g.lds_msgs, true, g.r_split, inplace) <= LDS_MAX) {
        g.lds_tables = true;
        g.lds = qbp::generic_lds_bytes(h->m, E1, h->n, g.lds_msgs, true, g.r_split, inplace);
    }
    g.threads = threads; g.per_cu = per_cu;
    g.grid = (int)std::max<long long>(1, std::min<long long>(B, (long long)h->num_cu * per_cu));
    return g;
}

static int generic_launch(qbp_handle* h, const uint8_t* d_syndromes, const double* d_prior, int64_t B,
                          int max_iter, int variant, double alpha, double damping, double clip_llr,
                          unsigned flags, uint8_t* d_hard, uint8_t* d_converged, int32_t* d_iters,
                          double* d_llr, double* d_dump, int dump_iter, double dump_div, hipStream_t s,
                          const qbp::GenericParams* mc = nullptr, int col_mode = 0)
{
    if ((flags & QBP_FLAG_PAIRWISE_COLSUM) && h->max_col_deg > qbp::GENERIC_PAIRWISE_MAX_COL)
        return fail(QBP_E_UNSUPPORTED, "QBP_FLAG_PAIRWISE_COLSUM supports column weights up to %d (got %d)",
                    qbp::GENERIC_PAIRWISE_MAX_COL, h->max_col_deg);
    // one launch hands out its syndromes through a 32-bit counter
    constexpr int64_t MAX_LAUNCH = (int64_t)1 << 30;
    if (B > MAX_LAUNCH) {
        if (!mc)
            return fail(QBP_E_UNSUPPORTED, "the general-H kernel decodes at most 2^30 syndromes per call (got %lld)",
                        (long long)B);
        for (int64_t off = 0; off < B; off += MAX_LAUNCH) {     // Monte-Carlo: trial ranges of any length
            qbp::GenericParams part = *mc;
            part.trial_begin += off;
            const int rc = generic_launch(h, nullptr, d_prior, std::min(MAX_LAUNCH, B - off), max_iter, variant,
                                          alpha, damping, clip_llr, flags, nullptr, nullptr, nullptr, nullptr,
                                          nullptr, 0, 1.0, s, &part);
            if (rc) return rc;
        }
        return QBP_OK;
    }
    const size_t E = (size_t)std::max(h->E, 1), n = (size_t)h->n;
    const bool inplace = QBP_GENERIC_INPLACE != 0 && variant == QBP_SUM_PRODUCT;
    const GenericGeom g = generic_geometry(h, B, inplace);
    if (g.lds > (size_t)160 * 1024)
        return fail(QBP_E_UNSUPPORTED, "m = %d checks need %zu B of LDS for the parity bits (limit 160 KiB)",
                    h->m, g.lds);
    if (!g.lds_msgs) {
        if (!inplace) HIP_TRY(h->d_wsQ.reserve((size_t)g.grid * E));
        HIP_TRY(h->d_wsR.reserve((size_t)g.grid * E));
    }
    if (inplace) HIP_TRY(h->d_wsV.reserve((size_t)g.grid * n));
    HIP_TRY(h->d_prior_sorted.reserve(n));
    HIP_TRY(qbp::launch_permute_prior(d_prior, h->d_svar.p, h->d_prior_sorted.p, (int)n, s));
    qbp::GenericParams G{};
    G.m = h->m; G.n = h->n; G.E = h->E;
    G.srow = h->d_srow.p; G.srow_e0 = h->d_srow_e0.p; G.srow_deg = h->d_srow_deg.p;
    G.col_idx = h->d_col_idx.p; G.epos = h->d_epos.p;
    G.long_edge_row = h->d_long_edge_row.p;
    {
        const size_t n_long = (size_t)(h->row_off[qbp::GENERIC_MAX_ROW_CLASS + 2] - h->row_off[qbp::GENERIC_MAX_ROW_CLASS + 1]);
        HIP_TRY(h->d_wsL.reserve((size_t)g.grid * 3 * std::max<size_t>(n_long, 1)));
        G.wsL = h->d_wsL.p;
    }
    std::copy(std::begin(h->row_off), std::end(h->row_off), G.row_off);
    std::copy(std::begin(h->row_base), std::end(h->row_base), G.row_base);
    G.svar = h->d_svar.p; G.lcol_ptr = h->d_lcol_ptr.p;
    G.vpos = col_mode == 1 ? h->f_order.vpos.p : h->d_vpos.p;
    G.vrow = col_mode == 1 ? h->f_order.vrow.p : h->d_vrow.p;
    G.vpos0 = col_mode == 2 ? h->f_order.vpos.p : nullptr;      // Fortran order at iteration 0 only
    G.vrow0 = col_mode == 2 ? h->f_order.vrow.p : nullptr;
    std::copy(std::begin(h->col_off), std::end(h->col_off), G.col_off);
    std::copy(std::begin(h->gcol_base), std::end(h->gcol_base), G.col_base);
    std::copy(std::begin(h->rpad_off), std::end(h->rpad_off), G.rpad_off);
    std::copy(std::begin(h->cpad_off), std::end(h->cpad_off), G.cpad_off);
    G.prior_sorted = h->d_prior_sorted.p;
    G.lds_tables = g.lds_tables ? 1 : 0;
    G.r_split = g.r_split;
    G.syndromes = d_syndromes; G.B = B; G.max_iter = max_iter; G.flags = flags;
    G.alpha = alpha; G.damping = damping; G.clip_llr = clip_llr;
    G.hard = d_hard; G.converged = d_converged; G.iters = d_iters; G.llr = d_llr;
    G.wsQ = h->d_wsQ.p; G.wsR = h->d_wsR.p; G.wsV = h->d_wsV.p;
    G.work_counter = h->d_work_counter.p;
    if (B > (int64_t)g.grid)     // (else every index it can yield is >= B whatever it holds)
        HIP_TRY(hipMemsetAsync(h->d_work_counter.p, 0, sizeof(unsigned long long), s));
    G.dump_R = d_dump; G.dump_iter = dump_iter; G.dump_div = dump_div;
    h->last_threads = g.threads; h->last_lds = (int)g.lds; h->last_grid = g.grid;
    if (mc) {
        // Monte-Carlo mode: per-workgroup scratch for the sampled error
        HIP_TRY(h->d_wsE.reserve((size_t)g.grid * ((n + 3) / 4) * 4));
        G.lx_cols = mc->lx_cols; G.trial_begin = mc->trial_begin; G.seed = mc->seed;
        G.threshold = mc->threshold; G.draws = mc->draws; G.half_distance = mc->half_distance;
        G.counters = mc->counters; G.wsE = h->d_wsE.p;
        G.errors_in = mc->errors_in;
        G.fail_list = mc->fail_list; G.fail_count = mc->fail_count; G.fail_syn = mc->fail_syn;
        G.fail_llr = mc->fail_llr; G.fail_hard = mc->fail_hard; G.fail_err = mc->fail_err;
        HIP_TRY(qbp::launch_generic(true, g.mem, variant, G, g.grid, g.threads, g.lds, s));
        return QBP_OK;
    }
    HIP_TRY(qbp::launch_generic(false, g.mem, variant, G, g.grid, g.threads, g.lds, s));
    return QBP_OK;
}

extern "C" {

const char* qbp_last_error(void) { return g_err; }
const char* qbp_version(void) { return "qbp 0.1 (gfx950)"; }

int qbp_create(const int32_t* row_ptr, const int32_t* col_idx, int32_t m, int32_t n,
               int32_t device, qbp_handle** out)
try {
    if (!out) return fail(QBP_E_INVALID, "out is null");
    *out = nullptr;
    HostTables T;
    int rc0 = build_tables(row_ptr, col_idx, m, n, T);
    if (rc0) return rc0;
    const int E = row_ptr[m];
    int ndev = 0;
    hipError_t de = hipGetDeviceCount(&ndev);
    if (de != hipSuccess || ndev <= 0)
        return fail(QBP_E_NO_DEVICE, "no HIP device available (%s); libqbp has no CPU fallback",
                    de == hipSuccess ? "device count 0" : hipGetErrorString(de));
    if (device < 0 || device >= ndev)
        return fail(QBP_E_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    DeviceScope on_device(device);
    HIP_TRY(on_device.err);

    struct Guard {                       // destroys a half-built handle on every early exit
        qbp_handle* h;
        ~Guard() { if (h) qbp_destroy(h); }
    } guard{new qbp_handle()};
    qbp_handle* h = guard.h;
    h->device = device; h->m = m; h->n = n; h->E = E;
    h->row_ptr.assign(row_ptr, row_ptr + m + 1);
    h->col_idx.assign(col_idx, col_idx + E);
    h->max_row_deg = T.max_row;
    h->max_col_deg = T.max_col;
    h->dc = T.dc; h->dv = T.dv; h->fused_ok = T.fused_ok; h->padded = T.padded;
    h->n_iso = (int)T.iso.size();
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    h->num_cu = prop.multiProcessorCount;

    hipError_t e1 = hipSuccess;
    auto up = [&](auto& buf, const auto& vec) {
        if (e1 != hipSuccess) return;
        e1 = buf.reserve(vec.size());
        if (e1 == hipSuccess && !vec.empty())
            e1 = hipMemcpy(buf.p, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice);
    };
    if (T.fused_ok) {
        up(h->d_tab_var, T.tab_var);
        up(h->d_tab_nbr, T.tab_nbr);
        up(h->d_tab_writer, T.tab_writer);
    }
    up(h->d_iso, T.iso);
    up(h->d_col_ptr, T.col_ptr);
    up(h->d_col_edge, T.col_edge);
    up(h->d_epos, T.epos);
    up(h->d_cpos, T.cpos);
    up(h->d_long_edge_row, T.long_edge_row);
    up(h->d_vpos, T.vpos);
    up(h->d_vrow, T.vrow);
    up(h->d_lcol_ptr, T.lcol_ptr);
    std::copy(std::begin(T.gcol_base), std::end(T.gcol_base), h->gcol_base);
    {   // work items of the general-H kernel: every weight class padded to whole wavefronts
        int off = 0;
        for (int k = 1; k <= qbp::GENERIC_MAX_ROW_CLASS; ++k) {
            h->rpad_off[k] = off;
            off += (T.row_off[k + 1] - T.row_off[k] + 63) / 64 * 64;
        }
        h->rpad_off[qbp::GENERIC_MAX_ROW_CLASS + 1] = off;
        off = 0;
        for (int k = 1; k <= qbp::GENERIC_MAX_COL_CLASS; ++k) {
            h->cpad_off[k] = off;
            off += (T.col_off[k + 1] - T.col_off[k] + 63) / 64 * 64;
        }
        h->cpad_off[qbp::GENERIC_MAX_COL_CLASS + 1] = off;
    }
    std::copy(std::begin(T.row_base), std::end(T.row_base), h->row_base);
    up(h->d_srow, T.srow);
    up(h->d_srow_e0, T.srow_e0);
    up(h->d_srow_deg, T.srow_deg);
    up(h->d_svar, T.svar);
    up(h->d_sedge, T.sedge);
    std::copy(std::begin(T.row_off), std::end(T.row_off), h->row_off);
    std::copy(std::begin(T.col_off), std::end(T.col_off), h->col_off);
    std::copy(std::begin(T.col_edge_base), std::end(T.col_edge_base), h->col_edge_base);
    up(h->d_row_ptr, h->row_ptr);
    up(h->d_col_idx, h->col_idx);
    {   // OSD-0: bit-packed rows of H and its CSR
        const int W = (n + 31) / 32;
        int NP = 1;
        while (NP < n) NP <<= 1;
        const size_t lds = qbp::osd_lds_bytes(m, n, W, NP);
        h->osd_W = W; h->osd_NP = NP; h->osd_lds = (int)((lds + 15) & ~(size_t)15);
        h->osd_ok = lds <= 64 * 1024 && m <= 64 * 32;   // (a lane tracks its rows in a 32-bit mask)
        if (h->osd_ok) {
            std::vector<uint32_t> hbits((size_t)m * W, 0u);
            for (int c = 0; c < m; ++c)
                for (int e = row_ptr[c]; e < row_ptr[c + 1]; ++e)
                    hbits[(size_t)c * W + (col_idx[e] >> 5)] |= 1u << (col_idx[e] & 31);
            up(h->d_hbits, hbits);
            // rank of H over GF(2) (bit-packed Gaussian elimination on the host, once per code):
            // the elimination loop of the kernel can stop as soon as that many pivots are found
            std::vector<uint32_t> A = hbits;
            int rank = 0;
            for (int col = 0; col < n && rank < m; ++col) {
                const int wi = col >> 5;
                const uint32_t bit = 1u << (col & 31);
                int piv = -1;
                for (int r = rank; r < m; ++r) if (A[(size_t)r * W + wi] & bit) { piv = r; break; }
                if (piv < 0) continue;
                if (piv != rank) for (int w = 0; w < W; ++w) std::swap(A[(size_t)piv * W + w], A[(size_t)rank * W + w]);
                for (int r = rank + 1; r < m; ++r)
                    if (A[(size_t)r * W + wi] & bit)
                        for (int w = 0; w < W; ++w) A[(size_t)r * W + w] ^= A[(size_t)rank * W + w];
                ++rank;
            }
            h->osd_rank = rank;
        }
    }
    if (e1 == hipSuccess) e1 = h->d_work_counter.reserve(1);
    if (e1 == hipSuccess) e1 = hipMemset(h->d_work_counter.p, 0, sizeof(unsigned long long));
    if (e1 == hipSuccess) e1 = h->d_fail_count.reserve(1);
    if (e1 == hipSuccess) e1 = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e1 != hipSuccess) return fail(QBP_E_HIP, "device setup failed: %s", hipGetErrorString(e1));
    guard.h = nullptr;
    *out = h;
    return QBP_OK;
}
QBP_ABI_CATCH

int qbp_plan(const int32_t* row_ptr, const int32_t* col_idx, int32_t m, int32_t n, int32_t info[8],
             int32_t* tab_var, uint16_t* tab_nbr, uint32_t* tab_writer)
try {
    HostTables T;
    int rc = build_tables(row_ptr, col_idx, m, n, T);
    if (rc) return rc;
    if (info) {
        info[0] = T.fused_ok ? 1 : 2; info[1] = T.dc; info[2] = T.dv; info[3] = T.max_row;
        info[4] = T.max_col; info[5] = (int32_t)T.iso.size(); info[6] = T.padded ? 1 : 0;
        info[7] = T.fused_ok ? (int32_t)fused_lds_bytes(T.dc, m, n, 1) : 0;
    }
    if (T.fused_ok) {
        if (tab_var) std::memcpy(tab_var, T.tab_var.data(), T.tab_var.size() * sizeof(int32_t));
        if (tab_nbr) std::memcpy(tab_nbr, T.tab_nbr.data(), T.tab_nbr.size() * sizeof(uint16_t));
        if (tab_writer) std::memcpy(tab_writer, T.tab_writer.data(), T.tab_writer.size() * sizeof(uint32_t));
    }
    return QBP_OK;
}
QBP_ABI_CATCH

int qbp_column_order(const int32_t* row_ptr, const int32_t* col_idx, int32_t m, int32_t n, int32_t col_order,
                     int32_t* col_ptr, int32_t* col_edge)
try {
    HostTables T;
    int rc = build_tables(row_ptr, col_idx, m, n, T, col_order);
    if (rc) return rc;
    if (col_ptr) std::memcpy(col_ptr, T.col_ptr.data(), ((size_t)n + 1) * sizeof(int32_t));
    if (col_edge && row_ptr[m] > 0) std::memcpy(col_edge, T.col_edge.data(), (size_t)row_ptr[m] * sizeof(int32_t));
    return QBP_OK;
}
QBP_ABI_CATCH

void qbp_destroy(qbp_handle* h)
{
    if (!h) return;
    DeviceScope on_device(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    h->d_tab_var.release(); h->d_tab_nbr.release(); h->d_tab_writer.release(); h->d_iso.release();
    h->d_work_counter.release(); h->d_syn.release(); h->d_hard.release(); h->d_conv.release();
    h->d_iters.release(); h->d_llr.release(); h->d_prior.release();
    h->d_mathx.release(); h->d_mathy.release(); h->d_part.release(); h->d_edges.release(); h->d_hist.release(); h->d_lx_cols.release(); h->d_counters.release();
    if (h->pin_host) (void)hipHostFree(h->pin_host);
    for (int i = 0; i < 2; ++i) {
        if (h->stage[i]) (void)hipHostFree(h->stage[i]);
        if (h->stage_ev[i]) (void)hipEventDestroy(h->stage_ev[i]);
    }
    h->d_col_ptr.release(); h->d_col_edge.release(); h->d_wsQ.release(); h->d_wsR.release();
    h->d_wsV.release(); h->d_wsC.release(); h->d_wsS.release(); h->d_wsE.release(); h->d_svar.release(); h->d_sedge.release();
    h->d_srow.release(); h->d_srow_e0.release(); h->d_srow_deg.release();
    h->d_epos.release(); h->d_cpos.release(); h->d_long_edge_row.release(); h->d_wsL.release();
    h->d_vpos.release(); h->d_vrow.release(); h->d_lcol_ptr.release(); h->d_prior_sorted.release();
    h->d_hbits.release(); h->d_row_ptr.release(); h->d_col_idx.release(); h->d_sol.release();
    h->f_order.tab_nbr.release(); h->f_order.tab_writer.release(); h->f_order.col_edge.release();
    h->f_order.sedge.release(); h->f_order.vpos.release(); h->f_order.vrow.release();
    h->d_osd_At.release(); h->d_osd_piv.release(); h->d_osd_idx.release(); h->d_osd_sol.release();
    h->d_osd_posn.release(); h->d_osd_redo.release(); h->d_osd_next.release();
    h->d_osd_keys.release();
    h->d_fail_list.release(); h->d_fail_count.release(); h->d_fail_syn.release();
    h->d_fail_hard.release(); h->d_fail_err.release(); h->d_fail_llr.release();
    delete h;
}

static int stream_launch(qbp_handle* h, const uint8_t* d_syndromes, const double* d_prior, int64_t B,
                         int max_iter, int variant, double alpha, double damping, double clip_llr,
                         unsigned flags, uint8_t* d_hard, uint8_t* d_converged, int32_t* d_iters,
                         double* d_llr, hipStream_t s, bool f_order = false)
{
    // streaming kernel: one lane per syndrome, messages [edge][syndrome] in a global workspace;
    // long batches go through in chunks that keep the workspace under 16 GiB
    const size_t E = (size_t)std::max(h->E, 1), n = (size_t)h->n, m = (size_t)h->m;
    const size_t per_lane = 2 * E * sizeof(double) + n + m;
    long long Bc = (long long)std::min<unsigned long long>(((unsigned long long)16 << 30) / per_lane,
                                                          (unsigned long long)B);
    Bc = std::max<long long>(256, (Bc + 255) / 256 * 256);
    const size_t msg_doubles = (size_t)Bc * E;
    HIP_TRY(h->d_wsQ.reserve(msg_doubles));
    HIP_TRY(h->d_wsR.reserve(msg_doubles));
    HIP_TRY(h->d_wsC.reserve((size_t)Bc * n));
    HIP_TRY(h->d_wsS.reserve((size_t)Bc * m));
    qbp::StreamParams P{};
    P.m = h->m; P.n = h->n; P.E = h->E;
    P.syndromes = d_syndromes; P.B = B; P.Bc = Bc;
    P.max_iter = max_iter; P.flags = flags; P.alpha = alpha; P.damping = damping; P.clip_llr = clip_llr;
    P.hard = d_hard; P.converged = d_converged; P.iters = d_iters; P.llr = d_llr;
    P.Q = h->d_wsQ.p; P.R = h->d_wsR.p; P.cand = h->d_wsC.p; P.synT = h->d_wsS.p;
    std::copy(std::begin(h->row_off), std::end(h->row_off), P.row_off);
    std::copy(std::begin(h->col_off), std::end(h->col_off), P.col_off);
    std::copy(std::begin(h->col_edge_base), std::end(h->col_edge_base), P.col_edge_base);
    for (long long b0 = 0; b0 < B; b0 += Bc) {
        P.b0 = b0;
        const long long lanes = std::min<long long>(Bc, B - b0);
        const unsigned grid = (unsigned)((lanes + 255) / 256);
        h->last_threads = 256; h->last_lds = 0; h->last_grid = (int)grid;
        HIP_TRY(qbp::launch_stream(variant, grid, P, h->d_col_idx.p, h->d_col_ptr.p,
                                   f_order ? h->f_order.col_edge.p : h->d_col_edge.p, d_prior,
                                   h->d_srow.p, h->d_srow_e0.p, h->d_srow_deg.p, h->d_svar.p,
                                   f_order ? h->f_order.sedge.p : h->d_sedge.p, s));
    }
    return QBP_OK;
}

int qbp_decode_batch_device(qbp_handle* h, const uint8_t* d_syndromes, const double* d_prior,
                            int64_t B, int32_t max_iter, int32_t variant, double alpha,
                            double damping, double clip_llr, uint32_t flags, uint8_t* d_hard,
                            uint8_t* d_converged, int32_t* d_iters, double* d_llr, void* stream)
try {
    int rc = check_decode_args(h, B, max_iter, variant);
    if (rc) return rc;
    if (B == 0) return QBP_OK;
    if (!d_syndromes || !d_prior) return fail(QBP_E_INVALID, "null input pointer");
    if (B > (int64_t)1 << 40) return fail(QBP_E_INVALID, "B too large");
    DeviceScope on_device(h->device);
    HIP_TRY(on_device.err);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int col_mode = 0;                        // column sums in the order of a Fortran-ordered dense R
    rc = resolve_column_order(h, flags, nullptr, &col_mode);
    if (rc) return rc;
    const bool f_order = col_mode == 1;
    // kernel choice: the on-chip kernel when the matrix fits; otherwise one workgroup per syndrome
    // (general-H), except for small graphs in batches that fill the chip with one LANE per syndrome,
    // where the streaming kernel is ahead (tools/bench_generic.py: [[288,12,18]], 262144 syndromes:
    // 3.7e6 against 2.6e6 /s; larger graphs keep their messages in L2 / Infinity Cache under the
    // general-H kernel and win there at every batch size measured)
    int kernel = h->opt_kernel;
    if (h->opt_force_generic) kernel = 2;
    if (kernel == 1 && !h->fused_ok) return fail(QBP_E_UNSUPPORTED, "H does not fit the on-chip kernel");
    if (kernel == 0) kernel = h->fused_ok ? 1 : ((B >= 131072 && h->E <= 2048) ? 3 : 2);
    // numpy's pairwise column sums only differ from 8 entries per column on; those matrices never
    // fit the on-chip kernel, and only the general-H kernel implements that order
    if ((flags & QBP_FLAG_PAIRWISE_COLSUM) && h->max_col_deg >= 8) kernel = 2;
    if (col_mode == 2) kernel = 2;           // two column orders in one launch: the general-H kernel only
    h->last_kernel = kernel;
    if (kernel == 3)
        return stream_launch(h, d_syndromes, d_prior, B, max_iter, variant, alpha, damping, clip_llr, flags,
                             d_hard, d_converged, d_iters, d_llr, s, f_order);
    if (kernel == 2)
        return generic_launch(h, d_syndromes, d_prior, B, max_iter, variant, alpha, damping, clip_llr,
                              flags, d_hard, d_converged, d_iters, d_llr, nullptr, 0, 1.0, s, nullptr, col_mode);
    LaunchCfg cfg;
    rc = make_cfg(h, B, &cfg, (flags & QBP_FLAG_FORCE_FULL) != 0, false);
    if (rc) return rc;
    FusedParams P{};
    fill_static(h, P, cfg, f_order);
    P.syndromes = d_syndromes; P.prior = d_prior; P.B = B;
    P.max_iter = max_iter; P.flags = flags;
    P.alpha = alpha; P.damping = damping; P.clip_llr = clip_llr;
    P.hard = d_hard; P.converged = d_converged; P.iters = d_iters; P.llr = d_llr;
    // The work counter hands out syndromes beyond the first one of each slot; when every syndrome
    // is some slot's first one (small calls) its value is irrelevant -- any index it yields is
    // >= B -- and the memset node is skipped (3 us of a 30 us single-syndrome call).
    if (B > (long long)cfg.grid * cfg.S)
        HIP_TRY(hipMemsetAsync(h->d_work_counter.p, 0, sizeof(unsigned long long), s));
    HIP_TRY((flags & QBP_FLAG_FAST_MATH) ? qbp::launch_fused_fast_math(false, variant, P, cfg, s)
                                         : qbp::launch_fused(false, variant, P, cfg, s));
    return QBP_OK;
}
QBP_ABI_CATCH

int qbp_decode_batch(qbp_handle* h, const uint8_t* syndromes, const double* prior, int64_t B,
                     int32_t max_iter, int32_t variant, double alpha, double damping,
                     double clip_llr, uint32_t flags, uint8_t* hard, uint8_t* converged,
                     int32_t* iters, double* llr)
try {
    int rc = check_decode_args(h, B, max_iter, variant);
    if (rc) return rc;
    if (B == 0) return QBP_OK;
    if (!syndromes || !prior) return fail(QBP_E_INVALID, "null input pointer");
    for (int v = 0; v < h->n; ++v)
        if (prior[v] != prior[v]) return fail(QBP_E_INVALID, "prior[%d] is NaN (+-inf are legal)", v);
    if (flags & QBP_FLAG_DENSE_F_COLSUM_ITER0) {     // (the shortcut is judged on the host copy of the priors)
        int mode = 0;
        unsigned fl = flags;
        rc = resolve_column_order(h, fl, prior, &mode);
        if (rc) return rc;
        if (mode == 0) flags = fl;                   // iteration 0 cannot depend on the order: plain call
    }
    DeviceScope on_device(h->device);
    HIP_TRY(on_device.err);
    const size_t m = h->m, n = h->n, b = (size_t)B;
    {
        // Small calls (the reference's one-syndrome-per-call usage, paperResults.py:71): inputs and
        // outputs live in ONE pinned host buffer that the kernel reads and writes directly over
        // PCIe; per call: two host memcpys, a memset node, the launch and one stream sync
        // instead of six hipMemcpyAsync round trips.
        const size_t off_prior = (b * m + 7) & ~(size_t)7;
        const size_t off_llr = off_prior + n * 8;
        const size_t off_iters = off_llr + b * n * 8;
        const size_t off_hard = off_iters + ((b * 4 + 7) & ~(size_t)7);
        const size_t off_conv = off_hard + b * n;
        const size_t total = off_conv + b;
        if (total <= (size_t)256 * 1024) {
            if (h->pin_bytes < total) {
                if (h->pin_host) (void)hipHostFree(h->pin_host);
                h->pin_host = nullptr; h->pin_bytes = 0;
                const size_t want = std::max<size_t>(total, 64 * 1024);
                HIP_TRY(hipHostMalloc(&h->pin_host, want, hipHostMallocMapped));
                HIP_TRY(hipHostGetDevicePointer(&h->pin_dev, h->pin_host, 0));
                h->pin_bytes = want;
            }
            uint8_t* ph = static_cast<uint8_t*>(h->pin_host);
            uint8_t* pd = static_cast<uint8_t*>(h->pin_dev);
            std::memcpy(ph, syndromes, b * m);
            std::memcpy(ph + off_prior, prior, n * 8);
            hipStream_t s = h->stream;
            rc = qbp_decode_batch_device(
                h, pd, reinterpret_cast<const double*>(pd + off_prior), B, max_iter, variant, alpha, damping,
                clip_llr, flags, hard ? pd + off_hard : nullptr, converged ? pd + off_conv : nullptr,
                iters ? reinterpret_cast<int32_t*>(pd + off_iters) : nullptr,
                llr ? reinterpret_cast<double*>(pd + off_llr) : nullptr, s);
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(s));
            if (hard) std::memcpy(hard, ph + off_hard, b * n);
            if (converged) std::memcpy(converged, ph + off_conv, b);
            if (iters) std::memcpy(iters, ph + off_iters, b * 4);
            if (llr) std::memcpy(llr, ph + off_llr, b * n * 8);
            return QBP_OK;
        }
    }
    HIP_TRY(h->d_syn.reserve(b * m));
    HIP_TRY(h->d_prior.reserve(n));
    if (hard) HIP_TRY(h->d_hard.reserve(b * n));
    if (converged) HIP_TRY(h->d_conv.reserve(b));
    if (iters) HIP_TRY(h->d_iters.reserve(b));
    if (llr) HIP_TRY(h->d_llr.reserve(b * n));
    hipStream_t s = h->stream;
    HIP_TRY(hipMemcpyAsync(h->d_syn.p, syndromes, b * m, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_prior.p, prior, n * sizeof(double), hipMemcpyHostToDevice, s));
    rc = qbp_decode_batch_device(h, h->d_syn.p, h->d_prior.p, B, max_iter, variant, alpha, damping,
                                 clip_llr, flags, hard ? h->d_hard.p : nullptr,
                                 converged ? h->d_conv.p : nullptr, iters ? h->d_iters.p : nullptr,
                                 llr ? h->d_llr.p : nullptr, s);
    if (rc) return rc;
    // Outputs: device -> pinned staging (DMA at PCIe rate) -> the caller's arrays (memcpy), in chunks
    // of up to 32 MiB, the host copy of one chunk overlapping the DMA of the next.  (A direct copy
    // into pageable memory goes through the runtime's own small staging buffers at a fraction of
    // that rate.)  The two staging buffers are as large as the largest chunk any call of this handle has
    // needed, never more than 32 MiB each (they used to be 32 MiB from the first 256 KiB call on).
    auto copy_out = [&]() -> int {
        const size_t per = (hard ? n : 0) + (converged ? 1 : 0) + (iters ? 4 : 0) + (llr ? 8 * n : 0);
        constexpr size_t STAGE = (size_t)32 << 20;
        if (per == 0) return QBP_OK;
        // (pieces of at least 1 MiB, at least four of them above 4 MiB: the reference driver's 13 MB per
        // batch then overlap their DMA with the copy-out instead of running one after the other)
        const size_t quarter = std::max<size_t>((b + 3) / 4, ((size_t)1 << 20) / per + 1);
        const size_t chunk = std::max<size_t>(1, std::min<size_t>({b, STAGE / (per + 8), quarter}));
        const size_t need = std::min<size_t>(STAGE, chunk * (per + 8) + 64);
        if (h->stage_bytes < need) {
            for (int i = 0; i < 2; ++i) {
                if (h->stage[i]) (void)hipHostFree(h->stage[i]);
                h->stage[i] = nullptr;
            }
            h->stage_bytes = 0;
            const size_t want = std::min<size_t>(STAGE, std::max<size_t>(need, (size_t)1 << 20));
            for (int i = 0; i < 2; ++i) {
                HIP_TRY(hipHostMalloc(&h->stage[i], want + 64, hipHostMallocDefault));
                if (!h->stage_ev[i]) HIP_TRY(hipEventCreateWithFlags(&h->stage_ev[i], hipEventDisableTiming));
            }
            h->stage_bytes = want;
        }
        struct Piece { size_t b0, cnt; };
        Piece prev{0, 0};
        auto offsets = [&](size_t cnt, size_t& o_llr, size_t& o_it, size_t& o_hard, size_t& o_conv) {
            o_llr = 0;
            o_it = o_llr + (llr ? cnt * n * 8 : 0);
            o_hard = o_it + (iters ? ((cnt * 4 + 7) & ~(size_t)7) : 0);
            o_conv = o_hard + (hard ? cnt * n : 0);
        };
        auto drain = [&](const Piece& pc, int slot) -> int {
            if (pc.cnt == 0) return QBP_OK;
            HIP_TRY(hipEventSynchronize(h->stage_ev[slot]));
            const uint8_t* st = static_cast<const uint8_t*>(h->stage[slot]);
            size_t o_llr, o_it, o_hard, o_conv;
            offsets(pc.cnt, o_llr, o_it, o_hard, o_conv);
            if (llr) std::memcpy(llr + pc.b0 * n, st + o_llr, pc.cnt * n * 8);
            if (iters) std::memcpy(iters + pc.b0, st + o_it, pc.cnt * 4);
            if (hard) std::memcpy(hard + pc.b0 * n, st + o_hard, pc.cnt * n);
            if (converged) std::memcpy(converged + pc.b0, st + o_conv, pc.cnt);
            return QBP_OK;
        };
        int slot = 0;
        for (size_t b0 = 0; b0 < b; b0 += chunk, slot ^= 1) {
            const size_t cnt = std::min(chunk, b - b0);
            uint8_t* st = static_cast<uint8_t*>(h->stage[slot]);
            size_t o_llr, o_it, o_hard, o_conv;
            offsets(cnt, o_llr, o_it, o_hard, o_conv);
            if (llr) HIP_TRY(hipMemcpyAsync(st + o_llr, h->d_llr.p + b0 * n, cnt * n * 8, hipMemcpyDeviceToHost, s));
            if (iters) HIP_TRY(hipMemcpyAsync(st + o_it, h->d_iters.p + b0, cnt * 4, hipMemcpyDeviceToHost, s));
            if (hard) HIP_TRY(hipMemcpyAsync(st + o_hard, h->d_hard.p + b0 * n, cnt * n, hipMemcpyDeviceToHost, s));
            if (converged) HIP_TRY(hipMemcpyAsync(st + o_conv, h->d_conv.p + b0, cnt, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(h->stage_ev[slot], s));
            const int rc2 = drain(prev, slot ^ 1);          // the previous chunk, while this one is in flight
            if (rc2) return rc2;
            prev = Piece{b0, cnt};
        }
        return drain(prev, slot ^ 1);
    };
    rc = copy_out();
    if (rc) {
        // (copies into the staging buffers may still be in flight: the next call must not find them so)
        (void)hipStreamSynchronize(s);
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(s));
    return QBP_OK;
}
QBP_ABI_CATCH

int qbp_check_messages(qbp_handle* h, const uint8_t* syndromes, const double* prior, int64_t B,
                       int32_t variant, double alpha, double damping, double clip_llr,
                       int32_t iteration, uint32_t flags, double* messages)
try {
    int rc = check_decode_args(h, B, iteration + 1, variant);
    if (rc) return rc;
    int col_mode = 0;                    // (only the column-sum order bits of `flags` are honoured)
    flags &= QBP_FLAG_DENSE_F_COLSUM | QBP_FLAG_DENSE_F_COLSUM_ITER0 | QBP_FLAG_PAIRWISE_COLSUM;
    if (prior) { rc = resolve_column_order(h, flags, prior, &col_mode); if (rc) return rc; }
    if (variant == QBP_SUM_PRODUCT) {
        // plain sum-product = the damped update with alpha = damping = 1 and no LLR clip: the
        // caller's alpha / damping / clip_llr are ignored, as QBP_SUM_PRODUCT ignores them everywhere
        variant = QBP_DAMPED_SP;
        alpha = 1.0; damping = 1.0; clip_llr = __builtin_inf();
    }
    if (B == 0) return QBP_OK;
    if (!syndromes || !prior || !messages) return fail(QBP_E_INVALID, "null pointer");
    DeviceScope on_device(h->device);
    HIP_TRY(on_device.err);
    const size_t m = h->m, n = h->n, b = (size_t)B, E = (size_t)std::max(h->E, 1);
    HIP_TRY(h->d_syn.reserve(b * m));
    HIP_TRY(h->d_prior.reserve(n));
    HIP_TRY(h->d_llr.reserve(b * E));
    hipStream_t s = h->stream;
    HIP_TRY(hipMemcpyAsync(h->d_syn.p, syndromes, b * m, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_prior.p, prior, n * sizeof(double), hipMemcpyHostToDevice, s));
    // min-sum returns R_new / alpha (rework/decoding.py:58-59); damped SP the unscaled R (:168-169,
    // taken before R * alpha)
    const double div = variant == QBP_MIN_SUM ? alpha : 1.0;
    rc = generic_launch(h, h->d_syn.p, h->d_prior.p, B, iteration + 1, variant, alpha, damping, clip_llr,
                        flags | QBP_FLAG_FORCE_FULL /* no early exit before the dump iteration */, nullptr,
                        nullptr, nullptr, nullptr, h->d_llr.p, iteration, div, s, nullptr, col_mode);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(messages, h->d_llr.p, b * (size_t)h->E * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return QBP_OK;
}
QBP_ABI_CATCH

int qbp_message_histograms(qbp_handle* h, const uint8_t* syndromes, const uint8_t* errors, const double* prior,
                           int64_t B, int32_t variant, double alpha, double damping, double clip_llr,
                           int32_t iteration, uint32_t flags, int32_t bins, double* edges, int64_t* hist0,
                           int64_t* hist1)
try {
    int rc = check_decode_args(h, B, iteration + 1, variant);
    if (rc) return rc;
    if (!syndromes || !errors || !prior || !edges || !hist0 || !hist1) return fail(QBP_E_INVALID, "null pointer");
    int col_mode = 0;                    // (only the column-sum order bits of `flags` are honoured)
    flags &= QBP_FLAG_DENSE_F_COLSUM | QBP_FLAG_DENSE_F_COLSUM_ITER0 | QBP_FLAG_PAIRWISE_COLSUM;
    rc = resolve_column_order(h, flags, prior, &col_mode);
    if (rc) return rc;
    if (bins < 1 || bins > 4096) return fail(QBP_E_INVALID, "bins = %d out of range (1 .. 4096)", bins);
    if (B == 0 || h->E == 0) return fail(QBP_E_INVALID, "no messages to bin (B = %lld, E = %d)", (long long)B, h->E);
    if (variant == QBP_SUM_PRODUCT) { variant = QBP_DAMPED_SP; alpha = 1.0; damping = 1.0; clip_llr = __builtin_inf(); }
    DeviceScope on_device(h->device);
    HIP_TRY(on_device.err);
    const size_t m = h->m, n = h->n, b = (size_t)B, E = (size_t)h->E;
    HIP_TRY(h->d_syn.reserve(b * m));
    HIP_TRY(h->d_hard.reserve(b * n));               // the true error bits
    HIP_TRY(h->d_prior.reserve(n));
    HIP_TRY(h->d_llr.reserve(b * E));                // the messages, [B][E]
    hipStream_t s = h->stream;
    HIP_TRY(hipMemcpyAsync(h->d_syn.p, syndromes, b * m, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_hard.p, errors, b * n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_prior.p, prior, n * sizeof(double), hipMemcpyHostToDevice, s));
    const double div = variant == QBP_MIN_SUM ? alpha : 1.0;
    rc = generic_launch(h, h->d_syn.p, h->d_prior.p, B, iteration + 1, variant, alpha, damping, clip_llr,
                        flags | QBP_FLAG_FORCE_FULL, nullptr, nullptr, nullptr, nullptr, h->d_llr.p, iteration,
                        div, s, nullptr, col_mode);
    if (rc) return rc;
    // range of all messages (rework/Alvarado.py:41-44: the two classes share one range)
    const int grid = (int)std::min<size_t>(1024, (b * E + 255) / 256);
    HIP_TRY(h->d_part.reserve((size_t)2 * grid));
    HIP_TRY(qbp::launch_hist_minmax(grid, h->d_llr.p, (long long)(b * E), h->d_part.p, s));
    std::vector<double> part((size_t)2 * grid);
    HIP_TRY(hipMemcpyAsync(part.data(), h->d_part.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    double lo = __builtin_inf(), hi = -__builtin_inf();
    for (int i = 0; i < grid; ++i) { lo = std::min(lo, part[2 * i]); hi = std::max(hi, part[2 * i + 1]); }
    if (!(lo <= hi) || std::isinf(lo) || std::isinf(hi))
        return fail(QBP_E_INVALID, "message range [%g, %g] is not finite (np.histogram raises there too)", lo, hi);
    if (lo == hi) { lo -= 0.5; hi += 0.5; }          // np.histogram's rule for a degenerate range
    // np.linspace(lo, hi, bins + 1): start + i * step, the last edge exactly hi
    const double step = (hi - lo) / (double)bins;
    for (int i = 0; i < bins; ++i) edges[i] = (double)i * step + lo;
    edges[bins] = hi;
    HIP_TRY(h->d_edges.reserve((size_t)bins + 1));
    HIP_TRY(h->d_hist.reserve((size_t)2 * bins));
    HIP_TRY(hipMemcpyAsync(h->d_edges.p, edges, ((size_t)bins + 1) * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(h->d_hist.p, 0, (size_t)2 * bins * sizeof(unsigned long long), s));
    const size_t lds = (((size_t)2 * bins + 1) & ~(size_t)1) * 4 + ((size_t)bins + 1) * 8;
    HIP_TRY(qbp::launch_hist_bin(grid, lds, h->d_llr.p, h->d_hard.p, h->d_col_idx.p, (long long)B, (int)E, (int)n,
                                 h->d_edges.p, bins, h->d_hist.p, s));
    std::vector<unsigned long long> hist((size_t)2 * bins);
    HIP_TRY(hipMemcpyAsync(hist.data(), h->d_hist.p, hist.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < bins; ++i) { hist0[i] = (int64_t)hist[i]; hist1[i] = (int64_t)hist[bins + i]; }
    return QBP_OK;
}
QBP_ABI_CATCH

static int mc_prepare(qbp_handle* h, const uint8_t* Lx, int32_t k, hipStream_t s)
{
    if (k < 0 || k > 64) return fail(QBP_E_INVALID, "k = %d logical operators (need 0..64)", k);
    if (k > 0 && !Lx) return fail(QBP_E_INVALID, "Lx is null");
    const size_t n = h->n;
    if (h->lx_cache_k == k && h->lx_cache.size() == (size_t)k * n &&
        (k == 0 || std::memcmp(h->lx_cache.data(), Lx, (size_t)k * n) == 0))
        return QBP_OK;
    std::vector<unsigned long long> cols(n, 0ull);
    for (int l = 0; l < k; ++l)
        for (size_t v = 0; v < n; ++v)
            if (Lx[(size_t)l * n + v] & 1) cols[v] |= 1ull << l;
    HIP_TRY(h->d_lx_cols.reserve(n));
    HIP_TRY(hipMemcpyAsync(h->d_lx_cols.p, cols.data(), n * sizeof(unsigned long long),
                           hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));   // cols is a local
    h->lx_cache.assign(Lx, Lx + (size_t)k * n);
    h->lx_cache_k = k;
    return QBP_OK;
}

static unsigned mc_threshold(double p)
{
    double t = std::floor(p * 4294967296.0);
    if (!(t > 0.0)) t = 0.0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (unsigned)t;
}

// Rank of H over GF(2) and its bit-packed rows for matrices beyond the one-wavefront kernel: built on
// first use (64-bit Gaussian elimination on the host: about half a second for 2592 x 7776), not in
// qbp_create, which most users of such matrices never follow with an OSD call.
static int osd_big_prepare(qbp_handle* h)
{
    if (h->osd_big_ready) return QBP_OK;
    const int m = h->m, n = h->n, W = h->osd_W;
    std::vector<uint32_t> hbits((size_t)m * W, 0u);
    for (int c = 0; c < m; ++c)
        for (int e = h->row_ptr[c]; e < h->row_ptr[c + 1]; ++e)
            hbits[(size_t)c * W + (h->col_idx[e] >> 5)] |= 1u << (h->col_idx[e] & 31);
    HIP_TRY(h->d_hbits.reserve(hbits.size()));
    HIP_TRY(hipMemcpy(h->d_hbits.p, hbits.data(), hbits.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    const int W64 = (n + 63) / 64;
    std::vector<uint64_t> A((size_t)m * W64, 0ull);
    for (int c = 0; c < m; ++c)
        for (int e = h->row_ptr[c]; e < h->row_ptr[c + 1]; ++e)
            A[(size_t)c * W64 + (h->col_idx[e] >> 6)] |= 1ull << (h->col_idx[e] & 63);
    int rank = 0;
    for (int col = 0; col < n && rank < m; ++col) {
        const int wi = col >> 6;
        const uint64_t bit = 1ull << (col & 63);
        int piv = -1;
        for (int r = rank; r < m; ++r) if (A[(size_t)r * W64 + wi] & bit) { piv = r; break; }
        if (piv < 0) continue;
        if (piv != rank) for (int w = 0; w < W64; ++w) std::swap(A[(size_t)piv * W64 + w], A[(size_t)rank * W64 + w]);
        for (int r = rank + 1; r < m; ++r)
            if (A[(size_t)r * W64 + wi] & bit)
                for (int w = wi; w < W64; ++w) A[(size_t)r * W64 + w] ^= A[(size_t)rank * W64 + w];
        ++rank;
    }
    h->osd_rank = rank;
    h->osd_big_ready = true;
    return QBP_OK;
}

// The one-pivot-at-a-time kernel, which follows the reference's row swaps (qbp_osd.hpp, osd0_big_kernel): matrices
// beyond 8192 rows, QBP_OPT_OSD_BIG = 2, and the records a fast kernel found inconsistent.
static int osd_launch_swaps(qbp_handle* h, const qbp::OsdParams& O, long long max_grid, hipStream_t s)
{
    const size_t m = h->m, n = h->n, RS = (size_t)h->osd_W + 1, NP = (size_t)h->osd_NP;
    const int grid = (int)std::max<long long>(1, std::min<long long>(max_grid, (long long)h->num_cu * 2));
    qbp::OsdBigWorkspace Wk{};
    Wk.keys_in_lds = NP * 12 <= (size_t)96 * 1024 ? 1 : 0;
    HIP_TRY(h->d_osd_At.reserve((size_t)grid * RS * m));
    HIP_TRY(h->d_osd_piv.reserve((size_t)grid * m));
    HIP_TRY(h->d_osd_posn.reserve((size_t)grid * 2 * m));
    HIP_TRY(h->d_osd_sol.reserve((size_t)grid * n));
    if (!Wk.keys_in_lds) {
        HIP_TRY(h->d_osd_keys.reserve((size_t)grid * NP));
        HIP_TRY(h->d_osd_idx.reserve((size_t)grid * NP));
    }
    Wk.At = h->d_osd_At.p; Wk.pivcol = h->d_osd_piv.p; Wk.posn = h->d_osd_posn.p; Wk.sol = h->d_osd_sol.p;
    Wk.keys = h->d_osd_keys.p; Wk.idx = h->d_osd_idx.p;
    HIP_TRY(qbp::launch_osd_big((unsigned)grid, Wk.keys_in_lds ? NP * 12 : 0, O, Wk, s));
    return QBP_OK;
}

// `redo`: the caller's syndromes may lie outside the column space of H (anything but the Monte-Carlo loop, whose
// syndromes come from errors): the fast kernels then list the records whose sweep says so, and the kernel that
// follows the reference's row swaps recomputes them (an empty list costs one small launch).
static int osd_launch(qbp_handle* h, qbp::OsdParams& O, long long max_items, hipStream_t s, bool redo = false)
{
    O.m = h->m; O.n = h->n; O.W = h->osd_W; O.NP = h->osd_NP;
    O.row_ptr = h->d_row_ptr.p; O.col_idx = h->d_col_idx.p;
    if (!h->osd_ok) {
        // matrices whose rows do not fit 64 KiB of LDS (rank and 32-bit rows: built on first use)
        int rc = osd_big_prepare(h);
        if (rc) return rc;
    }
    O.rank = h->osd_rank; O.hbits = h->d_hbits.p;
    const size_t m = h->m, n = h->n, NP = (size_t)h->osd_NP;
    // -- matrices beyond the LDS limit: eight pivots at a time (qbp_osd.hpp, osd0_blocked_kernel)
    qbp::OsdBigWorkspace Wk{};
    const size_t wc_max = (n + 63) / 64 + 1;
    const size_t LDS_MAX = 160 * 1024 - 64;                          // (static __shared__ of the kernel: 64 bytes)
    const size_t qs = 8 * wc_max * 8;
    const size_t act = m * 4;                                        // (row, D) list of a block's updates
    const size_t want_table = std::min<size_t>(96 * 1024, 256 * wc_max * 8);
    size_t region0 = std::max(want_table, NP * 8), lds = region0 + qs + NP * 4 + ((n + 15) & ~(size_t)15) + act;
    Wk.keys_in_lds = 1;
    if (lds > LDS_MAX) { Wk.keys_in_lds = 0; region0 = want_table; lds = region0 + qs + act; }
    const bool blocked = !h->osd_ok && h->opt_osd_big != 2 && m <= 8192 && lds <= LDS_MAX && 2 * wc_max * 8 <= want_table;
    if (!h->osd_ok && !blocked) return osd_launch_swaps(h, O, max_items, s);
    if (redo) {
        HIP_TRY(h->d_osd_redo.reserve((size_t)max_items + 1));
        HIP_TRY(hipMemsetAsync(h->d_osd_redo.p, 0, 8, s));
        O.redo = h->d_osd_redo.p;
    }
    if (blocked) {
        const int grid = (int)std::max<long long>(1, std::min<long long>(max_items, (long long)h->num_cu));
        Wk.wc_max = (int)wc_max;
        // first sweep: the 2048 least reliable columns (a sweep ends where no unpivoted row has a syndrome
        // bit left -- after some 200 columns on the BP failures of the space-time matrices in the tests,
        // 1632 at most)
        Wk.k_first = h->opt_osd_big == 3 ? 24 : (int)std::min<size_t>(n, 2048);
        Wk.lds_act = (int)(lds - act);
        HIP_TRY(h->d_osd_next.reserve(1));
        HIP_TRY(hipMemsetAsync(h->d_osd_next.p, 0, 8, s));
        Wk.next = h->d_osd_next.p;
        Wk.lds_region0 = (int)region0; Wk.lds_table = (int)want_table;
        HIP_TRY(h->d_osd_At.reserve((size_t)grid * wc_max * m * 2));
        if (!Wk.keys_in_lds) {
            HIP_TRY(h->d_osd_sol.reserve((size_t)grid * n));
            HIP_TRY(h->d_osd_keys.reserve((size_t)grid * NP));
            HIP_TRY(h->d_osd_idx.reserve((size_t)grid * NP));
        }
        Wk.At = h->d_osd_At.p; Wk.sol = h->d_osd_sol.p; Wk.keys = h->d_osd_keys.p; Wk.idx = h->d_osd_idx.p;
        const int rpt = m <= 1024 ? 1 : m <= 2048 ? 2 : m <= 4096 ? 4 : 8;
        HIP_TRY(qbp::launch_osd_blocked(rpt, (unsigned)grid, lds, O, Wk, s));
    } else {
        const long long grid = std::max<long long>(1, std::min<long long>(max_items, (long long)h->num_cu * 32));
        HIP_TRY(qbp::launch_osd_small(h->osd_W + 1, (unsigned)grid, (size_t)h->osd_lds, O, s));
    }
    if (redo) {
        qbp::OsdParams R = O;
        R.redo = nullptr; R.count = 0;
        R.count_ptr = h->d_osd_redo.p; R.list = h->d_osd_redo.p + 1;
        return osd_launch_swaps(h, R, std::min<long long>(max_items, 32), s);
    }
    return QBP_OK;
}

int qbp_osd0_batch_device(qbp_handle* h, const uint8_t* d_syndromes, const double* d_llr,
                          const uint8_t* d_hard, int64_t B, uint8_t* d_solution, void* stream)
try {
    if (!h) return fail(QBP_E_INVALID, "null handle");
    if (B < 0) return fail(QBP_E_INVALID, "B must be >= 0");
    if (B == 0) return QBP_OK;
    if (!d_syndromes || !d_llr || !d_hard || !d_solution) return fail(QBP_E_INVALID, "null pointer");
    DeviceScope on_device(h->device);
    HIP_TRY(on_device.err);
    qbp::OsdParams O{};
    O.count = B; O.syndromes = d_syndromes; O.llr = d_llr; O.hard = d_hard; O.solution = d_solution;
    return osd_launch(h, O, B, static_cast<hipStream_t>(stream), true);
}
QBP_ABI_CATCH

int qbp_osd0_batch(qbp_handle* h, const uint8_t* syndromes, const double* llr, const uint8_t* hard,
                   int64_t B, uint8_t* solution)
try {
    if (!h) return fail(QBP_E_INVALID, "null handle");
    if (B < 0) return fail(QBP_E_INVALID, "B must be >= 0");
    if (B == 0) return QBP_OK;
    if (!syndromes || !llr || !hard || !solution) return fail(QBP_E_INVALID, "null pointer");
    DeviceScope on_device(h->device);
    HIP_TRY(on_device.err);
    const size_t m = h->m, n = h->n, b = (size_t)B;
    HIP_TRY(h->d_syn.reserve(b * m));
    HIP_TRY(h->d_llr.reserve(b * n));
    HIP_TRY(h->d_hard.reserve(b * n));
    HIP_TRY(h->d_sol.reserve(b * n));
    hipStream_t s = h->stream;
    HIP_TRY(hipMemcpyAsync(h->d_syn.p, syndromes, b * m, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_llr.p, llr, b * n * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_hard.p, hard, b * n, hipMemcpyHostToDevice, s));
    int rc = qbp_osd0_batch_device(h, h->d_syn.p, h->d_llr.p, h->d_hard.p, B, h->d_sol.p, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(solution, h->d_sol.p, b * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return QBP_OK;
}
QBP_ABI_CATCH

static int mc_run_impl(qbp_handle* h, const uint8_t* Lx_host, int32_t k, int32_t distance,
                       double p, int32_t draws, uint64_t seed, int64_t trial_begin,
                       int64_t trial_end, const uint8_t* d_errors_in, const double* d_prior, int32_t max_iter,
                       int32_t variant, double alpha, double damping, double clip_llr,
                       uint32_t flags, int64_t* d_counters, void* stream)
{
    const int64_t T = trial_end - trial_begin;
    int rc = check_decode_args(h, T, max_iter, variant);
    if (rc) return rc;
    if (trial_begin < 0) return fail(QBP_E_INVALID, "trial_begin must be >= 0");
    if (draws != 1 && draws != 2) return fail(QBP_E_INVALID, "draws must be 1 or 2 (got %d)", draws);
    if (!(p >= 0.0 && p <= 1.0)) return fail(QBP_E_INVALID, "p = %g out of [0, 1]", p);
    if (!d_prior || !d_counters) return fail(QBP_E_INVALID, "null pointer");
    if (T == 0) return QBP_OK;
    DeviceScope on_device(h->device);
    HIP_TRY(on_device.err);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = mc_prepare(h, Lx_host, k, s);
    if (rc) return rc;
    const bool osd = (flags & QBP_FLAG_OSD0) != 0;
    if (osd) {
        // per-trial records of the trials BP leaves unconverged (read by the OSD kernel)
        const size_t t = (size_t)T, m = h->m, n = h->n;
        if (T > QBP_MC_OSD_MAX_TRIALS || t * (m + 10 * n) > ((size_t)16 << 30))
            return fail(QBP_E_INVALID, "with QBP_FLAG_OSD0 a call covers at most %lld trials of this matrix "
                                       "(got %lld); split the range",
                        (long long)std::min<size_t>(QBP_MC_OSD_MAX_TRIALS, ((size_t)16 << 30) / (m + 10 * n)),
                        (long long)T);
        HIP_TRY(h->d_fail_list.reserve(t));
        HIP_TRY(h->d_fail_syn.reserve(t * m));
        HIP_TRY(h->d_fail_llr.reserve(t * n));
        HIP_TRY(h->d_fail_hard.reserve(t * n));
        HIP_TRY(h->d_fail_err.reserve(t * n));
        HIP_TRY(hipMemsetAsync(h->d_fail_count.p, 0, sizeof(unsigned long long), s));
    }
    auto osd_pass = [&]() -> int {
        // second kernel: OSD-0 + classification of the trials BP left unconverged; their number is
        // read from device memory by the kernel itself (no host round trip)
        qbp::OsdParams O{};
        O.count_ptr = reinterpret_cast<const long long*>(h->d_fail_count.p);
        O.list = h->d_fail_list.p;
        O.syndromes = h->d_fail_syn.p; O.llr = h->d_fail_llr.p; O.hard = h->d_fail_hard.p;
        O.errors = h->d_fail_err.p; O.lx_cols = h->d_lx_cols.p; O.half_distance = distance / 2;
        O.counters = reinterpret_cast<long long*>(d_counters);
        return osd_launch(h, O, T, s);
    };
    if (!h->fused_ok || h->opt_kernel == 2 || h->opt_force_generic) {
        // matrices beyond the on-chip kernel: the whole loop inside the general-H kernel
        qbp::GenericParams M{};
        M.lx_cols = h->d_lx_cols.p; M.trial_begin = trial_begin; M.seed = seed;
        M.threshold = mc_threshold(p); M.draws = draws; M.half_distance = distance / 2;
        M.counters = reinterpret_cast<long long*>(d_counters);
        M.errors_in = d_errors_in;
        if (osd) {
            M.fail_list = h->d_fail_list.p; M.fail_count = h->d_fail_count.p;
            M.fail_syn = h->d_fail_syn.p; M.fail_llr = h->d_fail_llr.p;
            M.fail_hard = h->d_fail_hard.p; M.fail_err = h->d_fail_err.p;
        }
        h->last_kernel = 2;
        rc = generic_launch(h, nullptr, d_prior, T, max_iter, variant, alpha, damping, clip_llr, flags,
                            nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1.0, s, &M);
        if (rc) return rc;
        return osd ? osd_pass() : QBP_OK;
    }
    LaunchCfg cfg;
    rc = make_cfg(h, T, &cfg, (flags & QBP_FLAG_FORCE_FULL) != 0, true);
    if (rc) return rc;
    FusedParams P{};
    fill_static(h, P, cfg);
    P.prior = d_prior; P.B = T; P.max_iter = max_iter; P.flags = flags;
    P.alpha = alpha; P.damping = damping; P.clip_llr = clip_llr;
    P.lx_cols = h->d_lx_cols.p; P.trial_begin = trial_begin; P.seed = seed;
    P.threshold = mc_threshold(p); P.draws = draws; P.half_distance = distance / 2;
    P.counters = reinterpret_cast<long long*>(d_counters);
    P.errors_in = d_errors_in;
    if (osd) {
        P.fail_list = h->d_fail_list.p; P.fail_count = h->d_fail_count.p;
        P.fail_syn = h->d_fail_syn.p; P.fail_llr = h->d_fail_llr.p;
        P.fail_hard = h->d_fail_hard.p; P.fail_err = h->d_fail_err.p;
    }
    HIP_TRY(hipMemsetAsync(h->d_work_counter.p, 0, sizeof(unsigned long long), s));
    HIP_TRY((flags & QBP_FLAG_FAST_MATH) ? qbp::launch_fused_fast_math(true, variant, P, cfg, s)
                                         : qbp::launch_fused(true, variant, P, cfg, s));
    h->last_kernel = 1;
    if (osd) {
        rc = osd_pass();
        if (rc) return rc;
    }
    return QBP_OK;
}

int qbp_mc_run_device(qbp_handle* h, const uint8_t* Lx_host, int32_t k, int32_t distance,
                      double p, int32_t draws, uint64_t seed, int64_t trial_begin,
                      int64_t trial_end, const double* d_prior, int32_t max_iter,
                      int32_t variant, double alpha, double damping, double clip_llr,
                      uint32_t flags, int64_t* d_counters, void* stream)
try {
    return mc_run_impl(h, Lx_host, k, distance, p, draws, seed, trial_begin, trial_end, nullptr, d_prior, max_iter,
                       variant, alpha, damping, clip_llr, flags, d_counters, stream);
}
QBP_ABI_CATCH

int qbp_mc_run_errors(qbp_handle* h, const uint8_t* Lx, int32_t k, int32_t distance, const uint8_t* errors,
                      int64_t T, const double* prior, int32_t max_iter, int32_t variant, double alpha,
                      double damping, double clip_llr, uint32_t flags, int64_t counters[QBP_NUM_COUNTERS])
try {
    if (!h) return fail(QBP_E_INVALID, "null handle");
    if (!errors || !prior || !counters) return fail(QBP_E_INVALID, "null pointer");
    if (T < 0) return fail(QBP_E_INVALID, "T must be >= 0");
    for (int i = 0; i < QBP_NUM_COUNTERS; ++i) counters[i] = 0;
    if (T == 0) return QBP_OK;
    DeviceScope on_device(h->device);
    HIP_TRY(on_device.err);
    const size_t n = (size_t)h->n;
    HIP_TRY(h->d_prior.reserve(n));
    HIP_TRY(h->d_counters.reserve(QBP_NUM_COUNTERS));
    HIP_TRY(h->d_hard.reserve((size_t)T * n));            // (scratch of the host-pointer entries: the errors)
    hipStream_t s = h->stream;
    HIP_TRY(hipMemcpyAsync(h->d_prior.p, prior, n * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(h->d_hard.p, errors, (size_t)T * n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(h->d_counters.p, 0, QBP_NUM_COUNTERS * sizeof(long long), s));
    // (p, draws, seed are unused with stored errors)
    const int rc = mc_run_impl(h, Lx, k, distance, 0.0, 1, 0, 0, T, h->d_hard.p, h->d_prior.p, max_iter, variant,
                               alpha, damping, clip_llr, flags, reinterpret_cast<int64_t*>(h->d_counters.p), s);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    HIP_TRY(hipMemcpyAsync(counters, h->d_counters.p, QBP_NUM_COUNTERS * sizeof(long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return QBP_OK;
}
QBP_ABI_CATCH

int qbp_mc_run(qbp_handle* h, const uint8_t* Lx, int32_t k, int32_t distance, double p,
               int32_t draws, uint64_t seed, int64_t trial_begin, int64_t trial_end,
               const double* prior, int32_t max_iter, int32_t variant, double alpha,
               double damping, double clip_llr, uint32_t flags, int64_t counters[QBP_NUM_COUNTERS])
try {
    if (!h) return fail(QBP_E_INVALID, "null handle");
    if (!prior || !counters) return fail(QBP_E_INVALID, "null pointer");
    DeviceScope on_device(h->device);
    HIP_TRY(on_device.err);
    hipStream_t s = h->stream;
    HIP_TRY(h->d_prior.reserve(h->n));
    HIP_TRY(h->d_counters.reserve(qbp::NUM_COUNTERS));
    HIP_TRY(hipMemcpyAsync(h->d_prior.p, prior, h->n * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(h->d_counters.p, 0, qbp::NUM_COUNTERS * sizeof(long long), s));
    int rc = qbp_mc_run_device(h, Lx, k, distance, p, draws, seed, trial_begin, trial_end,
                               h->d_prior.p, max_iter, variant, alpha, damping, clip_llr, flags,
                               reinterpret_cast<int64_t*>(h->d_counters.p), s);
    if (rc) return rc;
    long long tmp[qbp::NUM_COUNTERS];
    HIP_TRY(hipMemcpyAsync(tmp, h->d_counters.p, sizeof(tmp), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < qbp::NUM_COUNTERS; ++i) counters[i] += tmp[i];
    return QBP_OK;
}
QBP_ABI_CATCH

int qbp_mc_sample_errors(qbp_handle* h, double p, int32_t draws, uint64_t seed,
                         int64_t trial_begin, int64_t T, uint8_t* errors)
try {
    if (!h) return fail(QBP_E_INVALID, "null handle");
    if (T < 0 || trial_begin < 0) return fail(QBP_E_INVALID, "T and trial_begin must be >= 0");
    if (!errors) return fail(QBP_E_INVALID, "errors is null");
    if (draws != 1 && draws != 2) return fail(QBP_E_INVALID, "draws must be 1 or 2");
    if (!(p >= 0.0 && p <= 1.0)) return fail(QBP_E_INVALID, "p = %g out of [0, 1]", p);
    if (T == 0) return QBP_OK;
    if (T > ((int64_t)1 << 31)) return fail(QBP_E_INVALID, "at most 2^31 trials per call");
    DeviceScope on_device(h->device);
    HIP_TRY(on_device.err);
    hipStream_t s = h->stream;
    const size_t n = h->n;
    // (the sampler does not depend on H: any matrix, whichever kernel decodes it)
    HIP_TRY(h->d_hard.reserve((size_t)T * n));
    HIP_TRY(qbp::launch_mc_sample(h->d_hard.p, h->n, T, trial_begin, draws, seed, mc_threshold(p), s));
    HIP_TRY(hipMemcpyAsync(errors, h->d_hard.p, (size_t)T * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return QBP_OK;
}
QBP_ABI_CATCH

int qbp_set_option(qbp_handle* h, int32_t option, int64_t value)
try {
    if (option == QBP_OPT_DEBUG_THROW) {          // tests: what an exception inside an entry point turns into
        if (value == 1) throw std::bad_alloc();
        if (value == 2) throw std::runtime_error("requested by QBP_OPT_DEBUG_THROW");
        if (value == 3) throw 3;
        return QBP_OK;
    }
    if (!h) return fail(QBP_E_INVALID, "null handle");
    switch (option) {
        case QBP_OPT_SLOTS_PER_BLOCK:
            if (value < 0 || value * h->m > qbp::FUSED_MAX_THREADS) return fail(QBP_E_INVALID, "slots*m must be <= %d", qbp::FUSED_MAX_THREADS);
            h->opt_slots = (int)value; return QBP_OK;
        case QBP_OPT_BLOCKS_PER_CU:
            if (value < 0 || value > 32) return fail(QBP_E_INVALID, "blocks per CU out of range");
            h->opt_blocks_per_cu = (int)value; return QBP_OK;
        case QBP_OPT_FORCE_GENERIC:
            h->opt_force_generic = value != 0; return QBP_OK;
        case QBP_OPT_KERNEL:
            if (value < 0 || value > 3) return fail(QBP_E_INVALID, "kernel selector out of range");
            h->opt_kernel = (int)value; return QBP_OK;
        case QBP_OPT_EARLY_EXIT_FULL_WG:
            h->opt_early_exit_full_wg = value != 0; return QBP_OK;
        case QBP_OPT_NO_FIRST_STEP_TABLE:
            h->opt_no_r0_table = value != 0; return QBP_OK;
        case QBP_OPT_FORCED_TWO_BARRIERS:
            h->opt_forced_two_barriers = value != 0; return QBP_OK;
        case QBP_OPT_OSD_BIG:
            if (value < 0 || value > 3) return fail(QBP_E_INVALID, "OSD kernel selector out of range");
            h->opt_osd_big = (int)value;
            if (value != 0) { h->osd_ok = false; }
            else {
                const size_t lds = qbp::osd_lds_bytes(h->m, h->n, h->osd_W, h->osd_NP);
                h->osd_ok = lds <= 64 * 1024 && h->m <= 64 * 32;
                h->osd_big_ready = false;       // (osd_rank / d_hbits are shared: rebuild on next use)
            }
            return QBP_OK;
        case QBP_OPT_GENERAL_NO_LDS_TABLES:
            h->opt_no_lds_tables = value != 0; return QBP_OK;
        case QBP_OPT_GENERAL_NO_R_SPLIT:
            h->opt_no_r_split = value != 0; return QBP_OK;
        case QBP_OPT_GENERAL_MEM:
            if (value < 0 || value > 2) return fail(QBP_E_INVALID, "memory mode out of range");
            h->opt_mem = (int)value; return QBP_OK;
        case QBP_OPT_GENERAL_THREADS:
            if (value < 0 || value > 1024) return fail(QBP_E_INVALID, "threads per workgroup out of range");
            h->opt_threads = (int)value; return QBP_OK;
        default: return fail(QBP_E_INVALID, "unknown option %d", option);
    }
}
QBP_ABI_CATCH

int64_t qbp_get_info(qbp_handle* h, int32_t what)
try {
    if (!h) return -1;
    switch (what) {
        case QBP_INFO_M: return h->m;
        case QBP_INFO_N: return h->n;
        case QBP_INFO_EDGES: return h->E;
        case QBP_INFO_MAX_ROW_DEG: return h->max_row_deg;
        case QBP_INFO_MAX_COL_DEG: return h->max_col_deg;
        case QBP_INFO_KERNEL_KIND:   // the kernel a decode call would use (small batch), see also ..._LAST
            if (h->opt_force_generic) return 2;
            if (h->opt_kernel) return h->opt_kernel;
            return h->fused_ok ? 1 : 2;   // (auto picks 3 for small graphs in batches >= 131072)
        case QBP_INFO_LAST_KERNEL: return h->last_kernel;
        case QBP_INFO_ONE_BARRIER: return h->last_one_barrier;
        case QBP_INFO_THREADS: return h->last_threads;
        case QBP_INFO_LDS_BYTES: return h->last_lds;
        case QBP_INFO_GRID: return h->last_grid;
        case QBP_INFO_NUM_CU: return h->num_cu;
        default: return -1;
    }
}
catch (...) { return -1; }

int qbp_debug_math(qbp_handle* h, int32_t kind, const double* x, double* y, int64_t count)
try {
    if (!h || !x || !y || count < 0 || kind < 0 || kind > 5) return fail(QBP_E_INVALID, "bad arguments");
    if (count == 0) return QBP_OK;
    DeviceScope on_device(h->device);
    HIP_TRY(on_device.err);
    HIP_TRY(h->d_mathx.reserve((size_t)count));
    HIP_TRY(h->d_mathy.reserve((size_t)count));
    hipStream_t s = h->stream;
    HIP_TRY(hipMemcpyAsync(h->d_mathx.p, x, (size_t)count * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(qbp::launch_debug_math(kind, h->d_mathx.p, h->d_mathy.p, (long long)count, s));
    HIP_TRY(hipMemcpyAsync(y, h->d_mathy.p, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return QBP_OK;
}
QBP_ABI_CATCH

}  // extern "C"
