#!/usr/bin/env python3
"""Golden vectors for OSD-0, produced by RUNNING THE REAL REFERENCE (build container only):

    MPLBACKEND=Agg python tests/golden/make_golden_osd.py

  decoding/OSD.py:3            performOSD(H, syndrome, llr, hard)
  decoding/OSD_enhanced.py:5   performOSD_enhanced(..., order=0)   (must agree with performOSD)
Inputs: (llr, hard) of reference BP runs that did not converge (performBeliefPropagationFast,
maxIter 30, Bernoulli(0.08) errors, rng seed 20260128), plus random (llr, hard) pairs, plus
random reliabilities containing +inf, -inf and NaN.
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = os.environ.get("QLDPC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):
    from decoding.beliefPropagation import performBeliefPropagationFast
    from decoding.OSD import performOSD
    from decoding.OSD_enhanced import performOSD_enhanced


def main():
    files = {"steane": "steane", "72": "[[72, 12, 6]]", "144": "[[144, 12, 12]]",
             "288": "[[288, 12, 18]]"}
    out = {}
    for tag, fname in files.items():
        H = np.load(os.path.join(REF, "codes", f"{fname}.npz"))["Hx"]
        m, n = H.shape
        rng = np.random.default_rng(20260128)
        syn, llrs, hards, sols, kinds = [], [], [], [], []
        p = 0.08 if tag != "steane" else 0.3
        tries = 0
        while len(syn) < (24 if tag != "steane" else 8) and tries < 5000:
            tries += 1
            e = (rng.random(n) < p).astype(int)
            s = (e @ H.T) % 2
            hard, conv, llr = performBeliefPropagationFast(H, s, np.full(n, np.log((1 - p) / p)),
                                                           verbose=False, maxIter=30)
            if conv:
                continue
            syn.append(s); llrs.append(llr); hards.append(hard); kinds.append(0)
        for _ in range(8):      # random reliabilities, random hard decision, consistent syndrome
            e = (rng.random(n) < 0.1).astype(int)
            syn.append((e @ H.T) % 2)
            llrs.append(rng.normal(0, 5, n)); hards.append((rng.random(n) < 0.05).astype(np.int8))
            kinds.append(1)
        for _ in range(6):      # non-finite reliabilities: np.argsort(|llr|) puts inf, then NaN, last
            e = (rng.random(n) < 0.1).astype(int)
            l = rng.normal(0, 5, n)
            pos = rng.choice(n, 3, replace=False)
            l[pos[0]], l[pos[1]], l[pos[2]] = np.inf, -np.inf, np.nan
            syn.append((e @ H.T) % 2); llrs.append(l)
            hards.append((l < 0).astype(np.int8)); kinds.append(2)
        for s, l, h in zip(syn, llrs, hards):
            a = performOSD(H, s, l, h)
            b = performOSD_enhanced(H, s, l, h, order=0)
            assert np.array_equal(a, b)
            # any order: OSD_enhanced returns the OSD-0 solution as soon as it matches the syndrome
            assert np.array_equal(a, performOSD_enhanced(H, s, l, h, order=2, max_combinations=50))
            assert np.array_equal((a @ H.T) % 2, s)
            sols.append(a)
        out[f"{tag}/H"] = H.astype(np.uint8)
        out[f"{tag}/syndromes"] = np.array(syn, np.uint8)
        out[f"{tag}/llr"] = np.array(llrs, np.float64)
        out[f"{tag}/hard"] = np.array(hards, np.uint8)
        out[f"{tag}/solution"] = np.array(sols, np.uint8)
        out[f"{tag}/kind"] = np.array(kinds, np.uint8)
        ties = sum(len(np.unique(np.abs(l))) < n for l in llrs)
        print(tag, len(syn), "cases,", ties, "with tied |llr|")
    path = os.path.join(HERE, "osd.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
