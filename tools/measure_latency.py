#!/usr/bin/env python3
"""Per-call latency of the drop-in single-syndrome entry points (the reference's usage pattern in
paperResults.py: one call per trial)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import bp, codes, osd  # noqa: E402

for name in ("[[72, 12, 6]]", "[[288, 12, 18]]"):
    code = codes.load_code(name)
    H = code.Hx
    rng = np.random.default_rng(0)
    p = 0.01
    prior = [np.log((1 - p) / p)] * code.n
    errs = (rng.random((2000, code.n)) < p).astype(int)
    syns = (errs @ H.T) % 2
    bp.performBeliefPropagationFast(H, syns[0], prior, verbose=False)
    t0 = time.perf_counter()
    for s in syns:
        det, ok, llr = bp.performBeliefPropagationFast(H, s, prior, verbose=False, maxIter=50)
    dt = time.perf_counter() - t0
    print(f"{name}: performBeliefPropagationFast {dt / len(syns) * 1e6:.1f} us per call ({len(syns) / dt:.0f} calls/s)")
    dec = bp.decoder_for(H)
    pr = np.asarray(prior)
    s8 = syns.astype(np.uint8)
    t0 = time.perf_counter()
    for s in s8:
        dec.decode(s[None, :], pr, 50)
    dt = time.perf_counter() - t0
    print(f"{name}: Decoder.decode(B=1) {dt / len(syns) * 1e6:.1f} us per call")
    t0 = time.perf_counter()
    for s in syns[:500]:
        osd.performOSD(H, s, llr, det)
    dt = time.perf_counter() - t0
    print(f"{name}: performOSD {dt / 500 * 1e6:.1f} us per call")
    t0 = time.perf_counter()
    for _ in range(2000):
        bp.decoder_for(H)
    print(f"{name}: decoder_for (content-hash cache lookup) {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us")

# one decode per call of the reference's largest space-time matrix (spaceTime.py:4-18 with
# [[288,12,18]] over 18 cycles: 2592 x 7776), general-H kernel
from scipy.sparse import csr_matrix, eye, hstack, kron   # noqa: E402

H = codes.load_code("[[288, 12, 18]]").Hx
m, T = H.shape[0], 18
Hst = hstack([kron(eye(T, dtype=np.int64), csr_matrix(H)),
              eye(m * T, dtype=np.int64) + eye(m * T, k=-m, dtype=np.int64)]).tocsr()
dec = bp.decoder_for(Hst)
rng = np.random.default_rng(3)
for p in (0.002, 0.02):
    err = (rng.random((50, Hst.shape[1])) < p).astype(np.int64)
    syn = ((Hst @ err.T).T % 2).astype(np.uint8)
    pr = np.full(Hst.shape[1], np.log((1 - p) / p))
    dec.decode(syn[:1], pr, 50)
    t0 = time.perf_counter()
    its = [int(dec.decode(s[None, :], pr, 50)[2][0]) for s in syn]
    dt = time.perf_counter() - t0
    print(f"space-time 2592 x 7776, p={p}: Decoder.decode(B=1) {dt / len(syn) * 1e6:.0f} us per call, "
          f"mean {np.mean(its) + 1:.1f} iterations, kernel {dec.info('last_kernel')}, "
          f"{dec.info('threads')} threads")
