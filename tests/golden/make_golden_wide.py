#!/usr/bin/env python3
"""A WIDE golden set: many syndromes per code, compact outputs, from the REAL reference.

    MPLBACKEND=Agg python tests/golden/make_golden_wide.py

For [[72,12,6]], [[144,12,12]], [[288,12,18]] and p in {0.03, 0.06, 0.09}: 1500 syndromes each
through rework/decoding.py:77 performBeliefPropagationFast (4-tuple, maxIter 50).  Stored per
syndrome: the syndrome and the hard decision (bit-packed), converged, iteration, and the LLR sum
(a one-number check of the posterior on converged syndromes).  Purpose: measure how often ANY other
tanh/atanh implementation (the oracle's libm, the device's qbp_math.hpp) disagrees with the
reference's numpy on hard decision / iteration over thousands of syndromes, not hundreds.
"""
import importlib.util
import os

import numpy as np

REF = os.environ.get("QLDPC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_rework_decoding", os.path.join(REF, "rework", "decoding.py"))
rework = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rework)

N_PER = 1500
out = {}
for tag, fname in (("72", "[[72, 12, 6]]"), ("144", "[[144, 12, 12]]"), ("288", "[[288, 12, 18]]")):
    H = np.load(os.path.join(REF, "codes", f"{fname}.npz"))["Hx"]
    m, n = H.shape
    rng = np.random.default_rng(20260129)
    for p in (0.03, 0.06, 0.09):
        prior = np.array([np.log((1 - p) / p)] * n)
        e = (rng.random((N_PER, n)) < p).astype(np.int64)
        syn = (e @ H.T) % 2
        hard = np.zeros((N_PER, n), np.uint8)
        conv = np.zeros(N_PER, np.uint8)
        iters = np.zeros(N_PER, np.int32)
        llr_sum = np.zeros(N_PER)
        for i in range(N_PER):
            h, c, l, it = rework.performBeliefPropagationFast(H, syn[i], prior, maxIter=50)
            hard[i], conv[i], iters[i], llr_sum[i] = h, c, it, l.sum()
        k = f"{tag}/p{p}"
        out[f"{k}/syndromes"] = np.packbits(syn.astype(np.uint8), axis=1)
        out[f"{k}/hard"] = np.packbits(hard, axis=1)
        out[f"{k}/converged"] = conv
        out[f"{k}/iters"] = iters
        out[f"{k}/llr_sum"] = llr_sum
        print(tag, p, "converged", int(conv.sum()), "/", N_PER, "mean iters", iters.mean() + 1)
# the other update rules on [[144,12,12]] (BASELINE config 3 parameterisation and the damped SP)
H = np.load(os.path.join(REF, "codes", "[[144, 12, 12]].npz"))["Hx"]
m, n = H.shape
rng = np.random.default_rng(20260130)
for vname, fn, kw in (("minsum", rework.performMinSum_Symmetric, dict(alpha=0.8, damping=0.7, clip_llr=25.0)),
                      ("sym", rework.performBeliefPropagation_Symmetric, dict(alpha=1.0, damping=0.8, clip_llr=20.0))):
    for p in (0.04, 0.08):
        prior = np.array([np.log((1 - p) / p)] * n)
        e = (rng.random((1000, n)) < p).astype(np.int64)
        syn = (e @ H.T) % 2
        hard = np.zeros((1000, n), np.uint8); conv = np.zeros(1000, np.uint8)
        iters = np.zeros(1000, np.int32); llr_sum = np.zeros(1000)
        for i in range(1000):
            h, c, l, it = fn(H, syn[i], prior, maxIter=50, **kw)
            hard[i], conv[i], iters[i], llr_sum[i] = h, c, it, l.sum()
        k = f"144{vname}/p{p}"
        out[f"{k}/syndromes"] = np.packbits(syn.astype(np.uint8), axis=1)
        out[f"{k}/hard"] = np.packbits(hard, axis=1)
        out[f"{k}/converged"] = conv
        out[f"{k}/iters"] = iters
        out[f"{k}/llr_sum"] = llr_sum
        print("144", vname, p, "converged", int(conv.sum()), "/ 1000")
path = os.path.join(HERE, "wide.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
