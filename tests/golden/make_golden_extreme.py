#!/usr/bin/env python3
"""Golden vectors for extreme priors, made by RUNNING THE REAL REFERENCE (build container only).

    MPLBACKEND=Agg python tests/golden/make_golden_extreme.py

Priors a caller can legally pass (main.py:18 computes log((1-p)/p) for whatever p it is given):
p = 0.5 (prior exactly 0: tanh = 0, the `|t| < 1e-15 -> 1e-15` branch of
decoding/beliefPropagation.py:122), p > 0.5 (negative priors), p = 0 and p = 1 (+-inf), very small p
(saturated tanh), and mixtures.  Same entry points and storage format as make_golden.py.
"""
import json
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import HERE, REF, SEED, run_single   # noqa: E402  (imports the reference)


def priors_for(n, rng):
    base = np.full(n, np.log(0.95 / 0.05))
    out = {}
    a = base.copy(); a[rng.choice(n, n // 6, replace=False)] = 0.0
    out["some priors exactly 0 (p = 0.5)"] = a
    out["all priors 0"] = np.zeros(n)
    a = base.copy(); a[rng.choice(n, n // 8, replace=False)] = -1.5
    out["some negative priors (p > 0.5)"] = a
    a = base.copy(); a[rng.choice(n, n // 6, replace=False)] = np.inf
    out["some priors +inf (p = 0)"] = a
    a = base.copy(); a[rng.choice(n, 3, replace=False)] = -np.inf
    out["some priors -inf (p = 1)"] = a
    out["p = 1e-12"] = np.full(n, np.log((1 - 1e-12) / 1e-12))
    out["p = 1e-300"] = np.full(n, np.log(1.0 / 1e-300))
    a = base.copy(); a[rng.choice(n, 6, replace=False)] = 1e-18
    a[rng.choice(n, 6, replace=False)] = -5e-16
    out["some |prior| < 2e-15 (the 1e-15 branch at iteration 0)"] = a
    a = rng.uniform(-3, 30, n); a[rng.choice(n, 4, replace=False)] = 0.0
    a[rng.choice(n, 4, replace=False)] = np.inf
    a[rng.choice(n, 4, replace=False)] = 1e-18
    a[rng.choice(n, 4, replace=False)] = -5e-16
    out["mixture incl. |prior| < 2e-15"] = a
    return out


def noise_driven(H, syn, prior, max_iter, damping=None, alpha=1.0, clip_llr=20.0):
    """Per-syndrome flag: after iteration 0 some edge had 0 < |Q| but |tanh(Q/2)| < 1e-15.

    Such a Q is the rounding residue of an exact cancellation (value - R); the reference then
    divides the row product by +1e-15 (beliefPropagation.py:122-123), which scales last-ulp noise
    of ITS tanh/arctanh to O(0.1) messages -- no other implementation can reproduce those outputs,
    so tests compare these syndromes structurally only.  Classification aid, not an expected
    output: an edge-list replay of the sum-product iteration (damping=None: a1/a3, else a5)."""
    rows, cols = np.nonzero(H)
    flags = np.zeros(len(syn), bool)
    with np.errstate(all="ignore"):
        for i, s in enumerate(syn):
            sign = 1.0 - 2.0 * s[rows]
            Q = prior[cols].astype(float)
            R = np.zeros(len(rows))
            for it in range(max_iter):
                t = np.tanh(Q / 2)
                # a tiny Q that is the difference of macroscopic numbers (not a tiny prior passed
                # through rows whose messages are exactly 0, which every implementation agrees on)
                if ((np.abs(t) < 1e-15) & (Q != 0) & (np.abs(R) > 1e-9)).any():
                    flags[i] = True
                    break
                prod = np.ones(H.shape[0])
                for e in range(len(rows)):                  # ascending column within a row
                    prod[rows[e]] *= t[e]
                ts = np.where(np.abs(t) < 1e-15, 1e-15, t)
                R = 2 * np.arctanh(np.clip(prod[rows] / ts * sign, -0.9999999, 0.9999999))
                if damping is not None:
                    R = R * alpha
                tot = np.zeros(H.shape[1])
                for e in range(len(rows)):                  # ascending check within a column
                    tot[cols[e]] += R[e]
                val = tot + prior
                Qn = val[cols] - R
                Q = Qn if damping is None else np.clip(damping * Qn + (1 - damping) * Q,
                                                       -clip_llr, clip_llr)
                if np.array_equal(((val < 0).astype(int) @ H.T) % 2, s):
                    break
    return flags


def main():
    warnings.simplefilter("ignore")
    H72 = np.load(os.path.join(REF, "codes", "[[72, 12, 6]].npz"))["Hx"]
    rng = np.random.default_rng(SEED + 1)
    Hr = np.zeros((24, 40), np.int64)
    for c in range(24):
        Hr[c, rng.choice(40, rng.integers(2, 10), replace=False)] = 1
    for tag, H in (("x72", H72), ("xrand", Hr)):
        m, n = H.shape
        arrays, manifest = {}, []
        e = (rng.random((13, n)) < 0.05).astype(np.int64)
        e[-3:] = rng.random((3, n)) < 0.25               # three heavy errors (rarely decodable)
        syn = np.concatenate([(e @ H.T) % 2, np.zeros((1, m), np.int64)])   # + the zero syndrome
        for note, prior in priors_for(n, rng).items():
            # 30 iterations, and 3 (where syndromes that never converge have not yet amplified the
            # last-ulp differences between tanh/arctanh implementations: compared value by value)
            for fn, mi, kw in (("fast4", 30, {}),
                               ("minsum", 30, dict(alpha=0.8, damping=0.7, clip_llr=25.0)),
                               ("sym", 30, {}),
                               ("fast4", 3, {}),
                               ("minsum", 3, dict(alpha=0.8, damping=0.7, clip_llr=25.0)),
                               ("sym", 3, {})):
                with np.errstate(all="ignore"):
                    hard, conv, iters, llr = run_single(fn, H, syn, prior, mi, **kw)
                k = f"case{len(manifest):02d}"
                arrays[f"{k}/syndromes"] = syn.astype(np.uint8)
                arrays[f"{k}/prior"] = prior
                arrays[f"{k}/hard"] = hard.astype(np.uint8)
                arrays[f"{k}/converged"] = conv.astype(np.uint8)
                arrays[f"{k}/iters"] = iters.astype(np.int32)
                arrays[f"{k}/llr"] = llr
                if fn == "minsum":
                    noisy = np.zeros(len(syn), bool)
                else:
                    noisy = noise_driven(H, syn, prior, mi, **({} if fn == "fast4" else
                                                               dict(damping=0.8, alpha=1.0, clip_llr=20.0)))
                arrays[f"{k}/noisy"] = noisy.astype(np.uint8)
                manifest.append(dict(key=k, fn=fn, max_iter=mi, note=note, kw=kw,
                                     n_converged=int(conv.sum()), B=int(len(conv))))
                print(f"  {tag} {k} {fn:6s} converged={int(conv.sum()):2d}/{len(conv)} "
                      f"nan={int(np.isnan(llr).sum())} inf={int(np.isinf(llr).sum())} "
                      f"noise-driven={int(noisy.sum())}  {note}")
        arrays["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
        arrays["H"] = H.astype(np.uint8)
        path = os.path.join(HERE, f"bp_{tag}.npz")
        np.savez_compressed(path, **arrays)
        print(f"wrote {path}: {os.path.getsize(path)} bytes, {len(manifest)} cases")


if __name__ == "__main__":
    main()
