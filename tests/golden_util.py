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
    """Yield dict cases: H, syndromes, prior, max_iter, variant, alpha, damping, clip_llr,
    and the reference's hard, converged, iters (-1 where the function returns none), llr."""
    d = np.load(os.path.join(GOLDEN, f"bp_{tag}.npz"))
    manifest = json.loads(bytes(d["manifest"]).decode())
    H = d["H"].astype(np.int64)
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


def compare(case, hard, conv, iters, llr, who):
    """Parity bar of BASELINE.json: hard decision, converged flag and iteration bit-exact;
    posterior LLR within 1e-5 relative on converged syndromes.  Non-converged syndromes ran
    max_iter chaotic iterations, where the reference's own tanh/arctanh (numpy SIMD) vs any
    other correctly-rounded-to-1ulp implementation drift apart (hard part 1):
    they get a loose bound and the worst drift is returned for reporting."""
    name = f"{who} {case['tag']}/{case['key']} {case['fn']} {case['note']}"
    assert np.array_equal(conv, case["converged"]), f"converged differs: {name}"
    assert np.array_equal(hard, case["hard"]), f"hard decision differs: {name}"
    if (case["iters"] >= 0).all():
        assert np.array_equal(iters, case["iters"]), f"iteration differs: {name}"
    else:   # 3-tuple reference functions: iteration not returned, check its invariants
        assert (iters[~conv] == case["max_iter"] - 1).all(), name
    ref = case["llr"]
    scale = np.maximum(np.abs(ref), 1e-300)
    rel = np.abs(llr - ref) / scale
    c = case["converged"]
    if c.any():
        assert rel[c].max() <= 1e-5, f"LLR rel err {rel[c].max():.3e} on converged: {name}"
    worst_nc = float(rel[~c].max()) if (~c).any() else 0.0
    if (~c).any():
        finite = np.isfinite(ref[~c]) & np.isfinite(llr[~c])
        assert np.array_equal(np.isfinite(ref[~c]), np.isfinite(llr[~c])), name
        # absolute drift bound relative to the message scale (|R| <= 16.81 per edge)
        assert np.abs(llr[~c] - ref[~c])[finite].max() <= 0.5, \
            f"LLR drift {np.abs(llr[~c] - ref[~c])[finite].max():.3e} on non-converged: {name}"
    return float(rel[c].max()) if c.any() else 0.0, worst_nc


# Extreme priors (tests/golden/make_golden_extreme.py): 0, negative, +-inf, saturating, mixtures
EXTREME_TAGS = ("x72", "xrand")


def compare_extreme(case, hard, conv, iters, llr, who):
    """Like `compare`, for vectors whose LLRs contain +-inf / NaN.  Converged flag and iteration:
    identical.  Syndromes that converged, and every syndrome of the 3-iteration cases (unless
    flagged noise-driven, below): non-finite pattern and hard decision identical, finite LLRs within 1e-5 relative or
    1e-7 absolute (saturated messages, |R| = 16.8 at the 0.9999999 clip, carry ~1e-9 of absolute
    noise from one ulp of the tanh product; a value that is a near-cancellation of such messages
    cannot be compared relatively).  Syndromes that ran 30 iterations without
    converging (random, undecodable syndromes are part of the set) are chaotic -- any two
    tanh/arctanh implementations drift apart -- and only counted."""
    name = f"{who} {case['tag']}/{case['key']} {case['fn']} {case['note']}"
    ref = case["llr"]
    assert np.array_equal(conv, case["converged"]), f"converged differs: {name}"
    assert np.array_equal(iters, case["iters"]), f"iteration differs: {name}"
    # `noisy` (set by the generator): the reference's own output is rounding noise scaled up.  A Q
    # that is the residue of an exact cancellation (|tanh| < 1e-15, Q != 0) makes
    # beliefPropagation.py:122-123 divide the row product by +1e-15, turning last-ulp differences
    # of ITS tanh/arctanh into O(0.1) messages; no other implementation reproduces those values.
    strict = case["converged"] & ~case["noisy"]
    if case["max_iter"] <= 3:
        strict = ~case["noisy"]
    r, x = ref[strict], llr[strict]
    assert np.array_equal(np.isnan(x), np.isnan(r)), f"NaN pattern differs: {name}"
    inf = np.isinf(r)
    assert np.array_equal(np.isinf(x), inf) and np.array_equal(x[inf], r[inf]), \
        f"inf pattern differs: {name}"
    fin = np.isfinite(r)
    diff = np.abs(np.where(fin, x, 0.0) - np.where(fin, r, 0.0))
    tol = np.maximum(1e-5 * np.abs(np.where(fin, r, 0.0)), 1e-7)
    assert (diff <= tol).all(), f"LLR err {diff.max():.3e}: {name}"
    # hard decision = (values < 0), False for NaN; sign noise around an exact cancellation aside
    solid = ~fin | (np.abs(np.where(fin, r, 1.0)) > 1e-7)
    assert np.array_equal(hard[strict][solid], case["hard"][strict][solid]), f"hard differs: {name}"
    loose = ~strict
    flips = int((hard[loose] != case["hard"][loose]).sum())
    return float(diff.max()) if diff.size else 0.0, flips
