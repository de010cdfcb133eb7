#!/usr/bin/env python3
"""Late-converging and non-converging syndromes with FULL posterior LLR vectors, from the REAL
reference -- the set that pins the LLR tolerance where it is hardest.

    MPLBACKEND=Agg python tests/golden/make_golden_late.py

For [[144,12,12]] and [[288,12,18]] at p in {0.05, 0.06}: candidate syndromes are drawn with a seeded
numpy Generator; the CPU oracle is used ONLY to pick the interesting ones quickly (those it decodes
after iteration 20, and a sample of those it does not decode in 50 iterations).  Every stored output
then comes from the reference itself:

* ``rework/decoding.py:77-129`` performBeliefPropagationFast (dense (m, n) form): hard decision,
  converged flag, iteration and the full ``values`` vector -> ``llr``;
* ``decoding/beliefPropagation.py:6-85`` performBeliefPropagation (its per-check / per-variable loop
  form: the same formula through numpy's scalar / short-array code paths): per-syndrome spread
  against the dense form (``self_rel`` = max_v |llr_loop - llr| / |llr|, ``self_abs``), and whether
  hard decision / converged flag agree (``self_same``).

Also stored: the spread of the same formula on glibc's tanh/atanh (the oracle) against the dense
form (``libm_rel``), so that three independent evaluations of the reference's formula bound what
"the reference's LLR" means on each syndrome.  The reference is not bit-stable against itself on
these inputs (hard part 1); tests/test_late_golden.py asserts device-vs-
reference drift per convergence-iteration bucket against that self-spread.
"""
import importlib.util
import io
import os
import sys
import contextlib

import numpy as np

REF = os.environ.get("QLDPC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
os.environ.setdefault("MPLBACKEND", "Agg")

spec = importlib.util.spec_from_file_location("ref_rework_decoding", os.path.join(REF, "rework", "decoding.py"))
rework = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rework)
from decoding.beliefPropagation import performBeliefPropagation  # noqa: E402  (the loop form)

from oracle import oracle  # noqa: E402  (candidate selection + the libm evaluation only)

N_CAND = int(os.environ.get("LATE_CANDIDATES", "40000"))
N_LATE = int(os.environ.get("LATE_PER_POINT", "160"))
N_NC = int(os.environ.get("LATE_NONCONV_PER_POINT", "60"))
MAX_ITER = 50


def rel_abs(a, b):
    d = np.abs(a - b)
    return float((d / np.maximum(np.abs(b), 1e-300)).max()), float(d.max())


out = {}
summary = []
for tag, fname in (("144", "[[144, 12, 12]]"), ("288", "[[288, 12, 18]]")):
    H = np.load(os.path.join(REF, "codes", f"{fname}.npz"))["Hx"]
    m, n = H.shape
    for p in (0.05, 0.06):
        rng = np.random.default_rng([20261004, n, int(p * 1000)])
        prior = np.array([np.log((1 - p) / p)] * n)
        e = (rng.random((N_CAND, n)) < p).astype(np.uint8)
        syn = (e.astype(np.int64) @ H.T % 2).astype(np.uint8)
        _, o_conv, o_it, o_llr = oracle.decode_batch(H, syn, prior, MAX_ITER, threads=8)
        late = np.flatnonzero(o_conv & (o_it > 20))
        # spread the late picks over the iteration range: the later, the rarer
        order = late[np.argsort(-o_it[late], kind="stable")]
        pick_late = np.sort(np.concatenate([order[: N_LATE // 2],
                                            rng.permutation(order[N_LATE // 2:])[: N_LATE - N_LATE // 2]]))
        pick_nc = np.flatnonzero(~o_conv)[:N_NC]
        pick = np.concatenate([pick_late, pick_nc])
        K = len(pick)
        hard = np.zeros((K, n), np.uint8); conv = np.zeros(K, np.uint8); iters = np.zeros(K, np.int32)
        llr = np.zeros((K, n)); self_rel = np.zeros(K); self_abs = np.zeros(K); self_same = np.zeros(K, np.uint8)
        libm_rel = np.zeros(K); libm_abs = np.zeros(K)
        for j, i in enumerate(pick):
            h, c, l, it = rework.performBeliefPropagationFast(H, syn[i], prior, maxIter=MAX_ITER)
            hard[j], conv[j], iters[j], llr[j] = h, c, it, l
            with contextlib.redirect_stdout(io.StringIO()):
                h2, c2, l2 = performBeliefPropagation(H, syn[i], prior, verbose=False, maxIter=MAX_ITER)
            self_rel[j], self_abs[j] = rel_abs(l2, l)
            self_same[j] = bool(c2 == c) and np.array_equal(h2, h)
            libm_rel[j], libm_abs[j] = rel_abs(o_llr[i], l)
            if j % 20 == 0:
                print(f"  {tag} p={p}: {j}/{K}", flush=True)
        k = f"{tag}/p{p}"
        out[f"{k}/syndromes"] = np.packbits(syn[pick], axis=1)
        out[f"{k}/hard"] = np.packbits(hard, axis=1)
        out[f"{k}/converged"] = conv
        out[f"{k}/iters"] = iters
        out[f"{k}/llr"] = llr
        out[f"{k}/self_rel"] = self_rel
        out[f"{k}/self_abs"] = self_abs
        out[f"{k}/self_same"] = self_same
        out[f"{k}/libm_rel"] = libm_rel
        out[f"{k}/libm_abs"] = libm_abs
        lc = conv.astype(bool) & (iters > 20)
        summary.append((tag, p, int(lc.sum()), int((~conv.astype(bool)).sum()),
                        int((iters[:len(pick_late)] != o_it[pick_late]).sum())))
        print(f"{tag} p={p}: candidates {N_CAND}, late convergers in the pool {len(late)}, stored "
              f"{int(lc.sum())} late + {int((~conv.astype(bool)).sum())} non-converged; "
              f"reference iteration != oracle iteration on {summary[-1][4]}; "
              f"loop form agrees on hard/converged: {int(self_same.sum())}/{K}", flush=True)
path = os.path.join(HERE, "late.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), summary)
