"""Loader for the committed golden vectors (tests/golden/bp_*.npz, made by make_golden.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAGS = ("steane", "72", "90", "108", "144", "288")
# matrices beyond the on-chip kernel: reference space-time matrix, random wide sparse matrix
IRREGULAR_TAGS = ("st72", "rand")
VARIANT = {"fast3": 0, "loop3": 0, "fast4": 0, "batch": 0, "sym": 1, "minsum": 2}
# argument defaults of the reference functions (rework/decoding.py:5 and :131)
DEFAULTS = {"minsum": dict(alpha=1.0, damping=1.0, clip_llr=20.0),
            "sym": dict(alpha=1.0, damping=0.8, clip_llr=20.0)}


def load(tag):
    """Yield dict cases: H (int64, in the memory order the reference was given), syndromes, prior, max_iter,
    variant, alpha, damping, clip_llr, and the reference's hard, converged, iters (-1 where the function
    returns none), llr."""
    d = np.load(os.path.join(GOLDEN, f"bp_{tag}.npz"))
    manifest = json.loads(bytes(d["manifest"]).decode())
    H = d["H"].astype(np.int64)             # (order 'K': keeps the Fortran order of the code files' Hx)
    for c in manifest:
        k = c["key"]
        kw = dict(alpha=1.0, damping=1.0, clip_llr=20.0)
        kw.update(DEFAULTS.get(c["fn"], {}))
        kw.update(c["kw"])
        yield dict(tag=tag, key=k, fn=c["fn"], note=c["note"], H=H, max_iter=c["max_iter"],
                   variant=VARIANT[c["fn"]], syndromes=d[f"{k}/syndromes"], prior=d[f"{k}/prior"],
                   hard=d[f"{k}/hard"], converged=d[f"{k}/converged"].astype(bool),
                   iters=d[f"{k}/iters"], llr=d[f"{k}/llr"],
                   errors=d[f"{k}/errors"] if f"{k}/errors" in d.files else None,
                   noisy=d[f"{k}/noisy"].astype(bool) if f"{k}/noisy" in d.files else None, **kw)


def oracle_flags(case):
    """Column-sum order of the reference function that produced the case (oracle flag bits)."""
    from oracle import oracle
    return oracle.colsum_flags(case["fn"], case["H"])


def device_flags(case):
    """The same as qbp_decode_batch flag bits -- what qldpc_amd's drop-in functions pass."""
    from qldpc_amd import _lib, bp
    fn = case["fn"]
    if fn == "loop3":
        return _lib.FLAG_PAIRWISE_COLSUM
    if fn == "batch":
        return 0
    return bp.dense_colsum_flags(case["H"], damped=fn in ("minsum", "sym"))


def same_bits(a, b):
    """Element-wise: identical float64 bit patterns (any NaN equals any NaN: payloads are not part of parity)."""
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))


def compare(case, hard, conv, iters, llr, who):
    """Round 3 parity bar: hard decision, converged flag, iteration AND posterior LLRs identical to the
    reference's, bit for bit, on every syndrome -- converged or not (BASELINE.json asks 1e-5 relative on
    the LLRs; oracle and device evaluate numpy's own tanh / arctanh kernels and add column sums in numpy's
    order, so there is no drift left to bound).  Returns the number of syndromes compared."""
    name = f"{who} {case['tag']}/{case['key']} {case['fn']} {case['note']}"
    assert np.array_equal(conv, case["converged"]), f"converged differs: {name}"
    assert np.array_equal(hard, case["hard"]), f"hard decision differs: {name}"
    if (case["iters"] >= 0).all():
        assert np.array_equal(iters, case["iters"]), f"iteration differs: {name}"
    else:   # 3-tuple reference functions: iteration not returned, check its invariants
        assert (iters[~conv] == case["max_iter"] - 1).all(), name
    same = same_bits(llr, case["llr"])
    if not same.all():
        rows = np.where(~same.all(1))[0]
        with np.errstate(all="ignore"):
            rel = np.abs(llr - case["llr"]) / np.maximum(np.abs(case["llr"]), 1e-300)
        raise AssertionError(f"LLR bits differ on {len(rows)} of {len(same)} syndromes (first {rows[:5]}, "
                             f"worst relative difference {np.nanmax(rel[~same]):.3e}): {name}")
    return len(same)


# Extreme priors (tests/golden/make_golden_extreme.py): 0, negative, +-inf, saturating, mixtures
EXTREME_TAGS = ("x72", "xrand")


def compare_extreme(case, hard, conv, iters, llr, who):
    """Vectors whose LLRs contain +-inf / NaN, exact cancellations included (the `noisy` syndromes, whose
    reference outputs are numpy's own last-ulp rounding scaled up by 1e15: beliefPropagation.py:122-123 --
    reproduced too, now that the arithmetic is numpy's): everything identical, as in `compare`."""
    return compare(case, hard, conv, iters, llr, who)
