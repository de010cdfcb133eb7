#!/usr/bin/env python3
"""The device-resident Monte-Carlo loop (sample -> decode -> OSD-0 on BP failures -> classify) on the 2592 x 7776
space-time matrix of [[288,12,18]] (general-H kernel + blocked OSD kernel); logical operators: the code's Lx on the
last cycle's data qubits (any 12 x n binary matrix serves for a rate measurement).  One JSON line."""
import json
import os
import sys
import time

import numpy as np
from scipy.sparse import csr_matrix

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes, mc  # noqa: E402

code = codes.load_code("[[288, 12, 18]]")
H, T = code.Hx, 18
m = H.shape[0]
Hst = np.hstack([np.kron(np.eye(T, dtype=np.int64), H),
                 (np.eye(m * T, dtype=np.int64) + np.eye(m * T, k=-m, dtype=np.int64)) % 2])
n = Hst.shape[1]
Lx = np.zeros((code.Lx.shape[0], n), np.uint8)
Lx[:, (T - 1) * code.n:T * code.n] = code.Lx
dec = _lib.Decoder(*bp.csr_from_H(csr_matrix(Hst)), bp.DEVICE)
out = {}
for p, trials in ((0.002, 40000), (0.005, 40000), (0.01, 20000)):
    prior = mc.prior_of(p, n)
    for flags, tag in ((0, "bp"), (_lib.FLAG_OSD0, "bp+osd0")):
        dec.mc_run(Lx, code.distance, p, prior, 0, 2000, seed=1, max_iter=50, flags=flags)     # tables, buffers
        t0 = time.perf_counter()
        c = dec.mc_run(Lx, code.distance, p, prior, 0, trials, seed=1, max_iter=50, flags=flags)
        dt = time.perf_counter() - t0
        out[f"p={p} {tag}"] = {"trials_per_s": round(trials / dt, 1), "not_converged": int(c[6]), "logical": int(c[1]),
                               "mean_iterations": round(float(c[7]) / trials + 1, 2), "osd_invalid": int(c[10])}
print(json.dumps(out))
