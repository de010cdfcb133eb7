#!/usr/bin/env python3
"""OSD-0 on syndromes OUTSIDE the column space of H, produced by RUNNING THE REAL REFERENCE (build container
only):    MPLBACKEND=Agg python tests/golden/make_golden_osd_inconsistent.py

  decoding/OSD.py:3   performOSD(H, syndrome, llr, hard)

No caller of the reference passes such a syndrome (every syndrome it decodes comes from an error), but the
function is defined on them: its elimination then ends with a 1 left in the syndrome column of a row without
a pivot, and the output depends on which ROW served as each column's pivot -- on the row swaps of
gf2_elimination (OSD.py:56-59).  Random syndromes (the Hx of the bivariate-bicycle codes have dependent rows:
almost every random syndrome is inconsistent), continuous random reliabilities (no ties), random hard decisions.
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
    from decoding.OSD import performOSD


def main():
    files = {"72": "[[72, 12, 6]]", "144": "[[144, 12, 12]]", "288": "[[288, 12, 18]]"}
    out = {}
    rng = np.random.default_rng(20261005)
    for tag, fname in files.items():
        H = np.load(os.path.join(REF, "codes", f"{fname}.npz"))["Hx"]
        m, n = H.shape
        syn, llrs, hards, sols, ok = [], [], [], [], []
        for _ in range(16):
            s = (rng.random(m) < 0.5).astype(int)
            l = rng.normal(0, 5, n)
            h = (rng.random(n) < 0.3).astype(np.int8)
            a = performOSD(H, s, l, h)
            syn.append(s); llrs.append(l); hards.append(h); sols.append(a)
            ok.append(bool(np.array_equal((a @ H.T) % 2, s)))
        out[f"{tag}/H"] = H.astype(np.uint8)
        out[f"{tag}/syndromes"] = np.array(syn, np.uint8)
        out[f"{tag}/llr"] = np.array(llrs, np.float64)
        out[f"{tag}/hard"] = np.array(hards, np.uint8)
        out[f"{tag}/solution"] = np.array(sols, np.uint8)
        out[f"{tag}/reproduces_syndrome"] = np.array(ok, np.uint8)
        print(tag, len(syn), "cases,", sum(ok), "of them consistent after all")
    path = os.path.join(HERE, "osd_inconsistent.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
