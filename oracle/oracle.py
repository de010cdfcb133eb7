"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module.  The product (``qldpc_amd``) never does.

The arithmetic lives in ``bp_oracle.c`` (each block cites the reference lines it restates);
this file only marshals arrays and restates the two integer-only helpers of the reference's
Monte-Carlo drivers in numpy:

* ``sample_errors_and_syndromes`` = decoding/beliefPropagationGPU.py:181-200
* ``classify_trials``             = paperResults_GPU.py:113-144 (= paperResults.py:83-100)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

VARIANT_SUM_PRODUCT = 0     # beliefPropagation.py:88-144 / rework/decoding.py:77-129
VARIANT_DAMPED_SP = 1       # rework/decoding.py:131-191
VARIANT_MIN_SUM = 2         # rework/decoding.py:5-75
FLAG_FORCE_FULL = 1
FLAG_PAIRWISE_COLSUM = 4   # column sums as np.sum(R[checks_v, v]) (loop form, beliefPropagation.py:68)
FLAG_LIBM_MATH = 8         # host libm tanh/atanh instead of numpy's own kernels (np_math.h)
FLAG_DENSE_F_COLSUM = 16   # column sums as np.sum(R, axis=0) on Fortran-ordered R (dense forms, F-ordered H)
FLAG_DENSE_F_COLSUM_ITER0 = 32   # ... at iteration 0 only (damped variants, F-ordered H below 256 KiB)


def colsum_flags(fn: str, H) -> int:
    """Order in which the reference function `fn` adds up a column of check->variable messages,
    given the caller's H -- numpy semantics, measured on numpy 2.2.6:

    * loop form (beliefPropagation.py:68): np.sum of the gathered column -> FLAG_PAIRWISE_COLSUM;
    * batch form (beliefPropagationGPU.py:147): C-ordered (B, m, n) temporaries -> row by row (0);
    * dense single-syndrome forms: row by row for a C-ordered H; for a Fortran-ordered H (the Hx of the
      reference's code files) every (m, n) temporary is F-ordered and np.sum(R, axis=0) is numpy's
      pairwise sum down the dense column -> FLAG_DENSE_F_COLSUM; the damped variants copy Q to C order
      (`Q_old = Q.copy()`), after which C order wins from iteration 1 on -> ..._ITER0, unless the arrays
      reach numpy's temporary-elision size (256 KiB), where the F-ordered temporary is reused.
    """
    if fn in ("loop3", "loop"):
        return FLAG_PAIRWISE_COLSUM
    if fn == "batch":
        return 0
    f_order = isinstance(H, np.ndarray) and H.ndim == 2 and H.flags["F_CONTIGUOUS"] and not H.flags["C_CONTIGUOUS"]
    if not f_order:
        return 0
    if fn in ("minsum", "sym") and H.shape[0] * H.shape[1] * 8 < 256 * 1024:
        return FLAG_DENSE_F_COLSUM_ITER0
    return FLAG_DENSE_F_COLSUM


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "bp_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_bp_decode_batch.restype = C.c_int
        L.oracle_bp_decode_batch.argtypes = [
            C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
            C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_uint32,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_bp_decode_batch_mt.restype = C.c_int
        L.oracle_bp_decode_batch_mt.argtypes = L.oracle_bp_decode_batch.argtypes + [C.c_int32]
        L.oracle_mc_errors.restype = None
        L.oracle_mc_errors.argtypes = [C.c_int32, C.c_double, C.c_int32, C.c_uint64, C.c_int64,
                                       C.c_int64, C.c_void_p]
        L.oracle_bp_check_messages.restype = C.c_int
        L.oracle_bp_check_messages.argtypes = [C.c_int32, C.c_int32] + [C.c_void_p] * 4 + [
            C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_uint32, C.c_void_p]
        L.oracle_osd0.restype = C.c_int
        L.oracle_osd0.argtypes = [C.c_int32, C.c_int32] + [C.c_void_p] * 5
        L.oracle_np_pairwise_sum.restype = C.c_double
        L.oracle_np_pairwise_sum.argtypes = [C.c_void_p, C.c_int32]
        for f in (L.oracle_np_tanh, L.oracle_np_arctanh):
            f.restype = None
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.oracle_mc_threshold.restype = C.c_uint32
        L.oracle_mc_threshold.argtypes = [C.c_double]
        _LIB = L
    return _LIB


def csr_of(H):
    """(row_ptr, col_idx) int32 with ascending columns per row, from dense or scipy-sparse H."""
    from scipy.sparse import csr_matrix, issparse
    S = csr_matrix(H) if issparse(H) else csr_matrix(np.asarray(H, dtype=np.float64))
    S.sort_indices()
    S.eliminate_zeros()
    return (np.ascontiguousarray(S.indptr, dtype=np.int32),
            np.ascontiguousarray(S.indices, dtype=np.int32), S.shape[0], S.shape[1])


def decode_batch(H, syndromes, prior, max_iter=50, variant=0, alpha=1.0, damping=1.0,
                 clip_llr=20.0, flags=0, threads=1):
    """Returns ``(hard u8[B,n], converged bool[B], iters i32[B], llr f64[B,n])``.
    ``threads`` > 1 spreads the (independent) syndromes over host threads (CPU baseline only)."""
    row_ptr, col_idx, m, n = csr_of(H)
    syn = np.ascontiguousarray(np.atleast_2d(np.asarray(syndromes)).astype(np.uint8))
    B = syn.shape[0]
    assert syn.shape[1] == m
    pr = np.ascontiguousarray(np.asarray(prior, dtype=np.float64))
    assert pr.shape == (n,)
    hard = np.zeros((B, n), np.uint8)
    conv = np.zeros(B, np.uint8)
    iters = np.zeros(B, np.int32)
    llr = np.zeros((B, n), np.float64)
    rc = lib().oracle_bp_decode_batch_mt(
        m, n, row_ptr.ctypes.data, col_idx.ctypes.data, syn.ctypes.data, pr.ctypes.data, B,
        int(max_iter), int(variant), float(alpha), float(damping), float(clip_llr), int(flags),
        hard.ctypes.data, conv.ctypes.data, iters.ctypes.data, llr.ctypes.data, int(threads))
    if rc != 0:
        raise ValueError(f"oracle_bp_decode_batch failed: {rc}")
    return hard, conv.astype(bool), iters, llr


def np_pairwise_sum(a):
    """The oracle's restatement of ``np.sum`` over a contiguous 1-D float64 array."""
    a = np.ascontiguousarray(a, np.float64)
    return float(lib().oracle_np_pairwise_sum(a.ctypes.data, len(a)))


def mc_errors(n, p, draws, seed, trial_begin, T):
    """The build's Philox error sampler (see bp_oracle.c), errors u8[T,n]."""
    out = np.zeros((T, n), np.uint8)
    lib().oracle_mc_errors(int(n), float(p), int(draws), int(seed), int(trial_begin), int(T),
                           out.ctypes.data)
    return out


def sample_errors_and_syndromes(H, error_rate, batch_size, rng):
    """decoding/beliefPropagationGPU.py:181-200 in numpy (host PCG64 stream)."""
    H = np.asarray(H)
    errors = (rng.random((batch_size, H.shape[1])) < error_rate).astype(np.int8)   # :195
    syndromes = (errors @ H.T) % 2                                                  # :198
    return errors, syndromes.astype(np.int8)


COUNTER_NAMES = ("trials", "logical_error", "BPs_fault", "BPs_miscorrected", "incorrectable",
                 "degenerateErrors", "not_converged", "sum_iterations",
                 "logical_error_not_converged", "exact_recoveries", "osd_invalid", "reserved1")


def classify_trials(H, Lx, distance, errors, syndromes, detections, converged, iters):
    """paperResults_GPU.py:113-144 without the OSD call (BP only): int64 counters[12]
    (layout of include/qbp.h: the reference's five counters plus bookkeeping)."""
    H = np.asarray(H).astype(np.int64)
    Lx = np.asarray(Lx).astype(np.int64)
    cnt = np.zeros(12, np.int64)
    for i in range(errors.shape[0]):
        error = errors[i].astype(np.int64)
        detection = detections[i].astype(np.int64)
        residual = (detection + error) % 2                                   # :127
        syndrome_logic = (Lx @ residual) % 2                                 # :129
        is_valid = np.array_equal((detection @ H.T) % 2, syndromes[i])       # :131-132
        if is_valid and not syndrome_logic.any() and not np.array_equal(detection, error):
            cnt[5] += 1                                                      # :134-135
        if syndrome_logic.any():
            cnt[1] += 1                                                      # :137-138
            if error.sum() < (distance // 2):                                # :140-141
                cnt[3] += 1
            else:
                cnt[4] += 1
            cnt[8] += int(not converged[i])
        cnt[9] += int(np.array_equal(detection, error))
        cnt[0] += 1
        cnt[6] += int(not converged[i])
        cnt[7] += int(iters[i])
    return cnt


def mc_counters(H, Lx, distance, p, prior, trial_begin, trial_end, draws=1, seed=0, max_iter=50,
                variant=0, alpha=1.0, damping=1.0, clip_llr=20.0, osd=False):
    """CPU statement of qbp_mc_run: Philox errors -> syndromes -> oracle decode [-> OSD-0 on the
    non-converged trials, paperResults.py:73-77] -> classification."""
    H = np.asarray(H).astype(np.int64)
    errors = mc_errors(H.shape[1], p, draws, seed, trial_begin, trial_end - trial_begin)
    syndromes = (errors.astype(np.int64) @ H.T % 2).astype(np.uint8)
    hard, conv, iters, llr = decode_batch(H, syndromes, prior, max_iter, variant, alpha, damping,
                                          clip_llr)
    if osd:
        hard = hard.copy()
        for i in np.flatnonzero(~conv):
            hard[i] = osd0(H, syndromes[i], llr[i], hard[i])
    cnt = classify_trials(H, Lx, distance, errors, syndromes, hard, conv, iters)
    if osd:     # invalid OSD outputs (never happens: the residual syndrome is in the column space)
        cnt[10] = sum(not np.array_equal((hard[i].astype(np.int64) @ H.T) % 2, syndromes[i])
                      for i in np.flatnonzero(~conv))
    return cnt


def osd0(H, syndrome, llr, hard):
    """decoding/OSD.py:3-28 performOSD (ties in |llr| broken by column index; see bp_oracle.c)."""
    Hb = np.ascontiguousarray(np.asarray(H) != 0, np.uint8)
    m, n = Hb.shape
    syn = np.ascontiguousarray(np.asarray(syndrome).astype(np.uint8) & 1)
    l = np.ascontiguousarray(llr, np.float64)
    h = np.ascontiguousarray(np.asarray(hard).astype(np.uint8) & 1)
    out = np.zeros(n, np.uint8)
    rank = lib().oracle_osd0(m, n, Hb.ctypes.data, syn.ctypes.data, l.ctypes.data, h.ctypes.data,
                             out.ctypes.data)
    if rank < 0:
        raise MemoryError
    return out


def check_messages(H, syndromes, prior, variant, alpha=1.0, damping=1.0, clip_llr=20.0, iteration=0, flags=0):
    """Edge messages float64[B, E] (CSR order): the reference's alpha_estimation=True output."""
    row_ptr, col_idx, m, n = csr_of(H)
    syn = np.ascontiguousarray(np.atleast_2d(np.asarray(syndromes)).astype(np.uint8))
    pr = np.ascontiguousarray(prior, np.float64)
    out = np.zeros((syn.shape[0], len(col_idx)), np.float64)
    rc = lib().oracle_bp_check_messages(m, n, row_ptr.ctypes.data, col_idx.ctypes.data,
                                        syn.ctypes.data, pr.ctypes.data, syn.shape[0], int(variant),
                                        float(alpha), float(damping), float(clip_llr),
                                        int(iteration), int(flags), out.ctypes.data)
    if rc:
        raise ValueError(rc)
    return out
