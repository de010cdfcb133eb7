/*
 * qbp.h -- C ABI of libqbp.so, the MI355X (gfx950) belief-propagation decoder.
 *
 * The reference (michelebanfi/qLDPC) has no FFI layer: its operator API for this path is a set
 * of Python functions.  Each entry point below names the reference function(s) it replaces;
 * qldpc_amd/bp.py and qldpc_amd/dropin/decoding/ mirror those Python signatures on top of
 * this ABI through ctypes (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every function returns 0 on success and a negative QBP_E_* code on failure; the message
 *     is available from qbp_last_error() (thread-local); no C++ exception crosses the ABI;
 *   - the caller owns every buffer; "host" entry points take host pointers, copy to and from
 *     the device on the handle's stream and synchronise before returning; "_device" entry
 *     points take device pointers, enqueue on the given HIP stream and return immediately;
 *   - a handle owns the device copies of the code's tables, is bound to one device and is not
 *     thread-safe (one handle per thread / GPU); distinct handles are independent;
 *   - launches of ONE handle share its work counter and scratch workspaces: "_device" calls on
 *     the same handle must be ordered on one stream (or by events); to overlap launches on
 *     several streams use one handle per stream;
 *   - every entry point makes the handle's device current for the duration of the call and
 *     restores the calling thread's device before returning;
 *   - there is NO CPU fallback: without a usable HIP device qbp_create fails with
 *     QBP_E_NO_DEVICE.
 */
#ifndef QBP_H
#define QBP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qbp_handle qbp_handle;

enum {
    QBP_OK = 0,
    QBP_E_INVALID = -1,     /* bad argument (shape, range, null pointer)            */
    QBP_E_NO_DEVICE = -2,   /* no HIP device / device index out of range            */
    QBP_E_HIP = -3,         /* a HIP runtime call failed                            */
    QBP_E_UNSUPPORTED = -4, /* operation not available for this matrix (Monte-Carlo / OSD limits) */
    QBP_E_NOMEM = -5
};

/* message-update rule */
enum {
    QBP_SUM_PRODUCT = 0, /* decoding/beliefPropagation.py:88-144 performBeliefPropagationFast
                            (= :6-85 performBeliefPropagation, rework/decoding.py:77-129,
                            decoding/beliefPropagationGPU.py:22-78 and :81-178 per sample) */
    QBP_DAMPED_SP = 1,   /* rework/decoding.py:131-191 performBeliefPropagation_Symmetric   */
    QBP_MIN_SUM = 2      /* rework/decoding.py:5-75    performMinSum_Symmetric              */
};

/* flags */
enum {
    QBP_FLAG_FORCE_FULL = 1u, /* run all max_iter iterations for every syndrome; outputs are still
                                 those of the first converged iteration (bench mode "M2") */
    QBP_FLAG_OSD0 = 2u,       /* qbp_mc_run only: trials BP does not converge on go through OSD-0
                                 (decoding/OSD.py) before classification, as paperResults.py:73-77 */
    QBP_FLAG_PAIRWISE_COLSUM = 4u, /* column sums in the order of np.sum over the gathered 1-D column,
                                 i.e. numpy's pairwise summation from 8 entries per column on: the
                                 loop form performBeliefPropagation (decoding/beliefPropagation.py:68).
                                 The dense forms (:129, rework/decoding.py) accumulate row by row =
                                 left to right, the default.  Only matrices with a column of weight
                                 >= 8 see a difference (they run on the general-H kernel). */
    QBP_FLAG_DENSE_F_COLSUM = 8u, /* column sums in the order of np.sum(R, axis=0) on a FORTRAN-ordered dense
                                 R: what the dense single-syndrome forms (decoding/beliefPropagation.py:129,
                                 rework/decoding.py:61,:119,:173) compute when the caller's H is Fortran-ordered,
                                 as the Hx of the reference's codes/ files is -- every (m, n) temporary inherits
                                 the layout of `mask = H != 0`, a column is contiguous, and numpy reduces it
                                 with its pairwise sum over all m entries (association by row index mod 8 and
                                 by halves above 128 rows).  The batch form (decoding/beliefPropagationGPU.py:147)
                                 works on C-ordered (B, m, n) arrays whatever H is: default order.
                                 QBP_E_UNSUPPORTED when some column's association is not a left-to-right sum
                                 of a reordering of its entries (possible from 4 entries per column on). */
    QBP_FLAG_FAST_MATH = 32u, /* opt-in, on-chip kernel only (matrices of the (6,3) / (8,4) shape classes; the other
                                 kernels ignore it): tanh and arctanh by rational / polynomial approximations of
                                 2.3 and 1.2 ulp (round 2's functions) instead of numpy's own two routines.  Posterior
                                 LLRs then follow numpy's to ~1e-6 relative while a syndrome converges early and
                                 drift apart like any other libm's on late convergers (tests/test_gpu_fast_math.py
                                 prints the table); hard decision, converged flag and iteration were identical
                                 to the reference's on every stored vector.  +28 % throughput on forced-50
                                 [[288,12,18]].  Default (flag clear): every output bit equal to the reference's. */
    QBP_FLAG_DENSE_F_COLSUM_ITER0 = 16u /* ... at iteration 0 only: the damped variants (rework/decoding.py:5,
                                 :131) on a Fortran-ordered H of fewer than 32768 entries -- `Q_old = Q.copy()`
                                 is C-ordered and C order wins in `damping * Q_new + (1 - damping) * Q_old` (:65,
                                 :179) from then on (from 256 KiB numpy reuses the F-ordered temporary instead:
                                 QBP_FLAG_DENSE_F_COLSUM).  Host-pointer entry point only; implemented for
                                 the case where iteration 0 cannot depend on the order (uniform priors, checks
                                 of equal weight, columns of at most 3 entries), else QBP_E_UNSUPPORTED. */
};
#define QBP_MC_OSD_MAX_TRIALS (1 << 20) /* per qbp_mc_run call with QBP_FLAG_OSD0 (record buffers) */

/*
 * Build a decoder for the parity-check matrix H given in CSR form.
 *   row_ptr [m+1], col_idx [E] (ascending within each row, no duplicates), m checks, n variables.
 * Replaces the per-call setup of the reference (csr_matrix(H), mask, adjacency lists:
 * decoding/beliefPropagation.py:12-23 and :93-101), done once per code instead of per syndrome.
 */
int qbp_create(const int32_t* row_ptr, const int32_t* col_idx, int32_t m, int32_t n,
               int32_t device, qbp_handle** out);
void qbp_destroy(qbp_handle* h);

/*
 * Host-only: what qbp_create would derive from the same CSR arrays, without touching a device
 * (used by the CPU test-suite).  info[8] = {kernel kind (1 on-chip, 2 general-H), DC, DV, max row
 * weight, max column weight, isolated variables, padded (0/1), LDS bytes of one slot}.  When the
 * on-chip kernel applies, the optional outputs receive its tables: tab_var [DC][m] (variable of
 * edge j of check c, -1 = padding), tab_nbr [DC][DV][m] (LDS word offsets j'*m + c' of the column
 * of that variable in ascending check order, DC*m = the zero word), tab_writer [m] (bit j: edge
 * (c, j) is the first of its column).
 */
int qbp_plan(const int32_t* row_ptr, const int32_t* col_idx, int32_t m, int32_t n, int32_t info[8],
             int32_t* tab_var, uint16_t* tab_nbr, uint32_t* tab_writer);

/*
 * Host-only: the order in which the kernels add up the check->variable messages of every column
 * (left to right).  col_order 0: ascending check -- np.sum(R, axis=0) on a C-ordered dense R
 * (decoding/beliefPropagation.py:129) adds row by row; col_order 1: the association numpy's pairwise sum
 * gives the column of a Fortran-ordered R (QBP_FLAG_DENSE_F_COLSUM).  Outputs: col_ptr [n+1] and col_edge [E]
 * (CSR edge ids, column by column, in summation order).  QBP_E_UNSUPPORTED when col_order 1 is not a
 * left-to-right sum for some column.
 */
int qbp_column_order(const int32_t* row_ptr, const int32_t* col_idx, int32_t m, int32_t n, int32_t col_order,
                     int32_t* col_ptr, int32_t* col_edge);

/*
 * Decode B syndromes (host buffers).
 *   syndromes [B][m] 0/1 bytes, prior [n] LLRs (initialBelief; +-inf allowed, NaN rejected with
 *   QBP_E_INVALID -- the _device entry points cannot check and clip NaN messages away), max_iter >= 1,
 *   variant QBP_*, alpha / damping / clip_llr as in rework/decoding.py (ignored by
 *   QBP_SUM_PRODUCT; alpha is R-scaling for QBP_DAMPED_SP and the normalisation for QBP_MIN_SUM).
 * Outputs (any may be NULL): hard [B][n] 0/1, converged [B] 0/1, iters [B] (0-based iteration of
 * the first syndrome match, max_iter-1 if none: rework/decoding.py:127,129), llr [B][n].
 * Replaces performBeliefPropagationBatch (decoding/beliefPropagationGPU.py:81-178) and, with
 * B = 1, every single-syndrome entry point listed under the variant enum.
 */
int qbp_decode_batch(qbp_handle* h, const uint8_t* syndromes, const double* prior, int64_t B,
                     int32_t max_iter, int32_t variant, double alpha, double damping,
                     double clip_llr, uint32_t flags, uint8_t* hard, uint8_t* converged,
                     int32_t* iters, double* llr);

/* Same, all pointers are DEVICE pointers, enqueued on `stream` (a hipStream_t, may be NULL),
 * asynchronous.  This is the call bench.py times with inputs resident in HBM. */
int qbp_decode_batch_device(qbp_handle* h, const uint8_t* d_syndromes, const double* d_prior,
                            int64_t B, int32_t max_iter, int32_t variant, double alpha,
                            double damping, double clip_llr, uint32_t flags, uint8_t* d_hard,
                            uint8_t* d_converged, int32_t* d_iters, double* d_llr, void* stream);

/*
 * Check->variable messages after the check update of iteration `iteration` (0-based), for B
 * syndromes: messages [B][E] in CSR edge order (host buffers).  This is the `alpha_estimation=True`
 * return value of the reference, restricted to the edges of H:
 *   QBP_MIN_SUM   rework/decoding.py:58-59   R_new / alpha at iteration 0
 *   QBP_DAMPED_SP rework/decoding.py:168-169 R (before scaling by alpha) at iteration 10
 * (QBP_SUM_PRODUCT: the plain update, i.e. alpha = damping = 1 and no LLR clip whatever is passed.)
 * flags: the column-sum order bits (QBP_FLAG_PAIRWISE_COLSUM / _DENSE_F_COLSUM / _DENSE_F_COLSUM_ITER0) of the
 * iterations before the dump; everything else is ignored.
 */
int qbp_check_messages(qbp_handle* h, const uint8_t* syndromes, const double* prior, int64_t B,
                       int32_t variant, double alpha, double damping, double clip_llr,
                       int32_t iteration, uint32_t flags, double* messages);

/*
 * The two histograms behind the alpha fit of rework/Alvarado.py:10-66 (estimate_alpha_from_code), on
 * the device: the check->variable messages of qbp_check_messages for B syndromes are binned by the
 * true value of the bit they address (errors [B][n] 0/1 bytes: class of message (b, e) =
 * errors[b][col_idx[e]], Alvarado.py:33-36) over their common range (:41-44) into `bins` equal bins
 * with np.histogram's rules (:46-47).  Outputs: edges [bins + 1] (= np.linspace(min, max, bins + 1)),
 * hist0 / hist1 [bins] raw counts.  The density normalisation and the one-parameter fit (:49-62)
 * are a few flops on 2 * bins numbers and stay with the caller (qldpc_amd/alvarado.py).
 */
int qbp_message_histograms(qbp_handle* h, const uint8_t* syndromes, const uint8_t* errors,
                           const double* prior, int64_t B, int32_t variant, double alpha,
                           double damping, double clip_llr, int32_t iteration, uint32_t flags, int32_t bins,
                           double* edges, int64_t* hist0, int64_t* hist1);

/*
 * Monte-Carlo trials [trial_begin, trial_end) entirely on the device: sample errors, form
 * syndromes, decode, classify, count.  Replaces the body of the trial loop of
 * paperResults_GPU.py:89-144 (= paperResults.py:57-100) without its OSD call:
 *   generate_errors_and_syndromes_batch (decoding/beliefPropagationGPU.py:181-200), `draws` = 2
 *   reproduces the XOR of two Bernoulli(p) draws (paperResults_GPU.py:96-105);
 *   performBeliefPropagationBatch; residual / logical check / counters (:113-144).
 * Errors come from Philox4x32-10 keyed by (seed, global trial index, qubit): the union of trials
 * is identical however the range is split over GPUs (oracle/bp_oracle.c states the sampler).
 *   Lx [k][n] 0/1 bytes (k <= 64), distance as in codes/<name>.npz, prior [n] (the decoder's
 *   prior is an input, as in the reference, and need not match p).
 * counters[QBP_NUM_COUNTERS] (int64, ADDED to):
 *   [0] trials  [1] logical_error  [2] BPs_fault (always 0, as in the reference)
 *   [3] BPs_miscorrected  [4] incorrectable  [5] degenerateErrors          (:80-84, :133-144)
 *   [6] not_converged (= trials the reference would hand to OSD)  [7] sum of iteration indices
 *   [8] logical_error among not_converged  [9] detection == error exactly
 *   [10] OSD outputs that miss the syndrome (always 0)  [11] reserved (0).
 * Any matrix: those that fit the on-chip kernel (m <= 1024, row weight <= 8, column weight <= 4) run
 * the fused loop, all others the Monte-Carlo mode of the general-H kernel; QBP_FLAG_OSD0 works with
 * both (matrices whose bit-packed rows exceed 64 KiB of LDS go through the workgroup-per-syndrome
 * OSD kernel, whose matrix copy lives in global memory).  With QBP_FLAG_OSD0 a call keeps per-trial
 * records (m + 10 n bytes each): at most QBP_MC_OSD_MAX_TRIALS trials and 16 GiB per call.
 */
#define QBP_NUM_COUNTERS 12
int qbp_mc_run(qbp_handle* h, const uint8_t* Lx, int32_t k, int32_t distance, double p,
               int32_t draws, uint64_t seed, int64_t trial_begin, int64_t trial_end,
               const double* prior, int32_t max_iter, int32_t variant, double alpha,
               double damping, double clip_llr, uint32_t flags, int64_t counters[QBP_NUM_COUNTERS]);

/* Asynchronous form: d_prior and d_counters (int64[QBP_NUM_COUNTERS], ADDED to) are device
 * pointers; Lx stays a host pointer (uploaded once per handle and cached). */
int qbp_mc_run_device(qbp_handle* h, const uint8_t* Lx_host, int32_t k, int32_t distance,
                      double p, int32_t draws, uint64_t seed, int64_t trial_begin,
                      int64_t trial_end, const double* d_prior, int32_t max_iter,
                      int32_t variant, double alpha, double damping, double clip_llr,
                      uint32_t flags, int64_t* d_counters, void* stream);

/*
 * The same pipeline -- syndrome = H e, BP, [OSD-0,] classification, all on the device -- on T GIVEN error
 * patterns (errors [T][n] 0/1 bytes, host) instead of sampled ones.  This is how the reference-pinned fixture of
 * the classification rule (tests/golden/classify.npz: trials sampled and classified by the reference's own
 * paperResults_GPU.py:113-144) is fed through the product path.  With QBP_FLAG_OSD0 at most
 * QBP_MC_OSD_MAX_TRIALS patterns per call.
 */
int qbp_mc_run_errors(qbp_handle* h, const uint8_t* Lx, int32_t k, int32_t distance, const uint8_t* errors,
                      int64_t T, const double* prior, int32_t max_iter, int32_t variant, double alpha,
                      double damping, double clip_llr, uint32_t flags, int64_t counters[QBP_NUM_COUNTERS]);

/*
 * OSD-0 post-processing of B decoder outputs: decoding/OSD.py:3-28 performOSD (= OSD_enhanced.py
 * with order 0), any matrix size.  syndromes [B][m], llr [B][n], hard [B][n] -> solution [B][n].  Columns are
 * ordered by ascending |llr|; equal values by ascending column index (np.argsort's order of
 * equal keys is unspecified in the reference).  A syndrome outside the column space of H gets the reference's
 * output too: there it depends on the row swaps of gf2_elimination (OSD.py:56-59); the records whose sweep
 * shows that are recomputed by a kernel that follows the swaps, in a second launch on the same stream.
 */
int qbp_osd0_batch(qbp_handle* h, const uint8_t* syndromes, const double* llr, const uint8_t* hard,
                   int64_t B, uint8_t* solution);
int qbp_osd0_batch_device(qbp_handle* h, const uint8_t* d_syndromes, const double* d_llr,
                          const uint8_t* d_hard, int64_t B, uint8_t* d_solution, void* stream);

/* Errors the sampler of qbp_mc_run draws for trials [trial_begin, trial_begin + T):
 * errors [T][n] host bytes.  For tests (compared bit for bit with the oracle's restatement). */
int qbp_mc_sample_errors(qbp_handle* h, double p, int32_t draws, uint64_t seed,
                         int64_t trial_begin, int64_t T, uint8_t* errors);

/* Tuning / introspection. */
enum {
    QBP_OPT_SLOTS_PER_BLOCK = 1, /* syndromes decoded concurrently by one workgroup (0 = auto) */
    QBP_OPT_BLOCKS_PER_CU = 2,   /* persistent workgroups per CU (0 = auto)                    */
    QBP_OPT_FORCE_GENERIC = 4,   /* 1 = use the general-H kernel even where the on-chip one fits */
    QBP_OPT_KERNEL = 5,          /* 0 auto, 1 on-chip, 2 general-H (workgroup per syndrome),
                                    3 streaming (lane per syndrome, messages in HBM)          */
    QBP_OPT_GENERAL_THREADS = 6, /* general-H kernel: threads per workgroup (0 = auto)        */
    QBP_OPT_GENERAL_MEM = 10,          /* general-H kernel, where the messages live: 0 auto (LDS when they fit),
                                          1 global workspace, 2 Q global + half of R in LDS (tests) */
    QBP_OPT_GENERAL_NO_R_SPLIT = 9,    /* 1 = general-H kernel keeps all of R in its global workspace (A/B) */
    QBP_OPT_GENERAL_NO_LDS_TABLES = 8, /* 1 = general-H kernel reads its variable-step tables from L2 (A/B) */
    QBP_OPT_EARLY_EXIT_FULL_WG = 13,  /* 1 = early-exit launches use workgroups of 16 wavefronts like forced ones
                                         (default: two of 8 per CU; A/B) */
    QBP_OPT_NO_FIRST_STEP_TABLE = 12, /* 1 = early-exit launches of the on-chip kernel compute a syndrome's first
                                         check step like every other (default: from a per-workgroup LDS table of
                                         its messages, which depend on the priors and the syndrome bit only; A/B, tests) */
    QBP_OPT_FORCED_TWO_BARRIERS = 11, /* 1 = QBP_FLAG_FORCE_FULL launches keep both barriers of the iteration
                                         (default: one barrier, two copies of the messages in LDS; A/B, tests) */
    QBP_OPT_OSD_BIG = 7,         /* tests: OSD-0 through a workgroup-per-syndrome kernel (matrix in global memory) even
                                    where the one-wavefront kernel fits.  1 = the kernel such matrices get (eight
                                    pivots at a time up to 8192 rows), 2 = the one-pivot-at-a-time kernel (larger
                                    matrices), 3 = as 1 with a first sweep over too few sorted columns, so that
                                    the full-width second sweep runs */
    QBP_OPT_DEBUG_THROW = 99,    /* tests (null handle allowed): raise inside the entry point -- 1 std::bad_alloc
                                    (-> QBP_E_NOMEM), 2 std::runtime_error, 3 a non-standard exception
                                    (-> QBP_E_INVALID): no exception crosses the ABI */
    QBP_INFO_M = 100, QBP_INFO_N = 101, QBP_INFO_EDGES = 102, QBP_INFO_MAX_ROW_DEG = 103,
    QBP_INFO_MAX_COL_DEG = 104, QBP_INFO_KERNEL_KIND = 105, /* 1 on-chip, 2 general-H, 3 streaming */
    QBP_INFO_THREADS = 106, QBP_INFO_LDS_BYTES = 107, QBP_INFO_GRID = 108, QBP_INFO_NUM_CU = 109,
    QBP_INFO_LAST_KERNEL = 110, /* kernel of the last decode launch: 1 on-chip, 2 general-H, 3 streaming */
    QBP_INFO_ONE_BARRIER = 111  /* 1 if the last on-chip launch geometry used the one-barrier forced kernel */
};
int qbp_set_option(qbp_handle* h, int32_t option, int64_t value);
int64_t qbp_get_info(qbp_handle* h, int32_t what);

/* Device evaluation of the kernels' FP64 elementary functions, for tests:
 * kind 0: np.tanh(x * 0.5), 1: 2.0 * np.arctanh(x) as the kernels compute them (numpy's bits: qbp_math.hpp),
 * 2: raw v_rcp_f64(x), 3: div_nr(1, x), 4 / 5: the round-1/2 forms of 0 / 1 (QBP_MATH_FAST builds).
 * Host buffers. */
int qbp_debug_math(qbp_handle* h, int32_t kind, const double* x, double* y, int64_t count);

const char* qbp_last_error(void);
const char* qbp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QBP_H */
