#!/usr/bin/env python3
"""Golden vectors on a matrix with HEAVY COLUMNS (>= 8 checks per variable), from the real reference.

    MPLBACKEND=Agg python tests/golden/make_golden_heavy.py

The reference's two sum-product forms part ways on such matrices: the loop form
(decoding/beliefPropagation.py:68) sums a gathered 1-D array with ``np.sum`` -- numpy's pairwise
order from 8 terms on -- while the dense form (:129, ``np.sum(R, axis=0)``) accumulates row by row.
Cases ``loop3`` are decoded with the loop form, ``fast3`` / ``fast4`` with the dense form, on the
same inputs.  Matrix: 160 x 56, column weights 1..40 plus one column of weight 136 (numpy's pairwise
sum recurses above 128 terms), an isolated variable, row weights up to 22.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import prior_of, run_single  # noqa: E402  (imports the reference)

rng = np.random.default_rng(20261004)
m, n = 160, 56
H = np.zeros((m, n), np.int64)
weights = [1, 2, 3, 7, 8, 9, 15, 16, 17, 24, 31, 32, 33, 40, 136] + list(rng.integers(4, 26, n - 16))
for v, w in enumerate(weights):
    H[rng.choice(m, int(w), replace=False), v] = 1          # column n-1 stays isolated
pv = rng.uniform(0.01, 0.12, n)
prior = np.log((1 - pv) / pv)
e = (rng.random((24, n)) < pv).astype(np.int64)
syn = (e @ H.T) % 2
arrays, manifest = {}, []
for fn, mi in (("loop3", 4), ("loop3", 30), ("fast4", 4), ("fast4", 30)):
    B = 12 if fn == "loop3" else 24
    hard, conv, iters, llr = run_single(fn, H, syn[:B], prior, mi)
    k = f"case{len(manifest):02d}"
    arrays[f"{k}/syndromes"] = syn[:B].astype(np.uint8)
    arrays[f"{k}/prior"] = prior
    arrays[f"{k}/hard"] = hard.astype(np.uint8)
    arrays[f"{k}/converged"] = conv.astype(np.uint8)
    arrays[f"{k}/iters"] = iters.astype(np.int32)
    arrays[f"{k}/llr"] = llr
    arrays[f"{k}/errors"] = e[:B].astype(np.uint8)
    manifest.append(dict(key=k, fn=fn, max_iter=mi, note="heavy columns", kw={},
                         n_converged=int(conv.sum()), B=B))
    print(f"  heavy {k} {fn} maxIter={mi} B={B} converged={int(conv.sum())} "
          f"col wt<= {int(H.sum(0).max())} row wt<= {int(H.sum(1).max())}")
# how far apart the reference's two forms are on these inputs (same syndromes, 4 iterations)
d = np.abs(arrays["case00/llr"] - arrays["case02/llr"][:12])
print("loop form vs dense form after 4 iterations: max |dLLR| =", d.max(),
      "entries that differ:", int((d > 0).sum()), "of", d.size)
arrays["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
arrays["H"] = H.astype(np.uint8)
path = os.path.join(HERE, "bp_heavy.npz")
np.savez_compressed(path, **arrays)
print("wrote", path, os.path.getsize(path))
