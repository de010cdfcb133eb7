#!/usr/bin/env python3
"""Golden vectors for the NaN path of the damped sum-product variant, from the REAL reference.

    MPLBACKEND=Agg python tests/golden/make_golden_damping1.py

rework/decoding.py:131 performBeliefPropagation_Symmetric with damping = 1.0 and an infinite prior:
Q = damping * Q_new + (1 - damping) * Q_old = 1 * inf + 0 * inf = NaN at iteration 0 (:179); numpy then
carries the NaN through tanh, the row product, np.clip and arctanh into every message of the rows
that touch such a variable, and `values < 0` is False for NaN.  Same storage format as
make_golden_extreme.py, including its `noisy` flag for the finite-prior cases.
"""
import json
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import HERE, REF, SEED, run_single   # noqa: E402  (imports the reference)
from make_golden_extreme import noise_driven          # noqa: E402  (classification aid, see there)

warnings.simplefilter("ignore")
H72 = np.load(os.path.join(REF, "codes", "[[72, 12, 6]].npz"))["Hx"]
rng = np.random.default_rng(SEED + 7)
Hr = np.zeros((24, 40), np.int64)
for c in range(24):
    Hr[c, rng.choice(40, rng.integers(2, 10), replace=False)] = 1
for tag, H in (("xd72", H72), ("xdrand", Hr)):
    m, n = H.shape
    arrays, manifest = {}, []
    e = (rng.random((10, n)) < 0.05).astype(np.int64)
    syn = np.concatenate([(e @ H.T) % 2, np.zeros((1, m), np.int64)])
    base = np.full(n, np.log(0.95 / 0.05))
    priors = {}
    a = base.copy(); a[rng.choice(n, 3, replace=False)] = np.inf
    priors["three priors +inf"] = a
    a = base.copy(); a[rng.choice(n, 2, replace=False)] = -np.inf; a[rng.choice(n, 2, replace=False)] = np.inf
    priors["priors +inf and -inf"] = a
    priors["finite priors (damping 1 = plain update with clip)"] = base.copy()
    for note, prior in priors.items():
        for mi, kw in ((3, dict(alpha=1.0, damping=1.0, clip_llr=20.0)),
                       (30, dict(alpha=1.0, damping=1.0, clip_llr=20.0)),
                       (30, dict(alpha=0.9, damping=1.0, clip_llr=25.0))):
            with np.errstate(all="ignore"):
                hard, conv, iters, llr = run_single("sym", H, syn, prior, mi, **kw)
            k = f"case{len(manifest):02d}"
            arrays[f"{k}/syndromes"] = syn.astype(np.uint8)
            arrays[f"{k}/prior"] = prior
            arrays[f"{k}/hard"] = hard.astype(np.uint8)
            arrays[f"{k}/converged"] = conv.astype(np.uint8)
            arrays[f"{k}/iters"] = iters.astype(np.int32)
            arrays[f"{k}/llr"] = llr
            # syndromes whose reference output is an exact cancellation scaled by 1e15 (finite priors only;
            # NaN trajectories are compared by their NaN / inf pattern, which no rounding can change)
            noisy = noise_driven(H, syn, prior, mi, damping=kw["damping"], alpha=kw["alpha"],
                                 clip_llr=kw["clip_llr"]) if np.isfinite(prior).all() else np.zeros(len(syn), bool)
            arrays[f"{k}/noisy"] = noisy.astype(np.uint8)
            manifest.append(dict(key=k, fn="sym", max_iter=mi, note=note, kw=kw,
                                 n_converged=int(conv.sum()), B=int(len(conv))))
            print(f"  {tag} {k} sym maxIter={mi} converged={int(conv.sum())}/{len(conv)} "
                  f"nan={int(np.isnan(llr).sum())} inf={int(np.isinf(llr).sum())}  {note} {kw}")
    arrays["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
    arrays["H"] = H.astype(np.uint8)
    path = os.path.join(HERE, f"bp_{tag}.npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {os.path.getsize(path)} bytes, {len(manifest)} cases")
