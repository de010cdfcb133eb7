#!/usr/bin/env python3
"""A/B numbers for the register-budget switches of qbp_kernels.hpp (library from QBP_LIB_PATH):
forced-50 rate of the on-chip (8, 4) kernel on the 864 x 2592 space-time matrix, the headline (6, 3)
rate, and the Monte-Carlo loop at p = 0.05 / 0.01 (+ OSD).  One line of JSON."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes  # noqa: E402

dev = torch.device("cuda", 0)
out = {}


def space_time(H, T):
    m = H.shape[0]
    return np.hstack([np.kron(np.eye(T, dtype=np.int64), H),
                      (np.eye(m * T, dtype=np.int64) + np.eye(m * T, k=-m, dtype=np.int64)) % 2])


def forced(H, B, p=0.01):
    mm, n = H.shape
    g = torch.Generator(device=dev); g.manual_seed(3)
    err = torch.rand((B, n), generator=g, device=dev) < p
    syn = (err.float() @ torch.from_numpy(H.T.astype(np.float32)).to(dev)).remainder_(2).to(torch.uint8)
    prior = torch.full((n,), float(np.log((1 - p) / p)), dtype=torch.float64, device=dev)
    hard = torch.empty((B, n), dtype=torch.uint8, device=dev); conv = torch.empty((B,), dtype=torch.uint8, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev); llr = torch.empty((B, n), dtype=torch.float64, device=dev)
    from scipy.sparse import csr_matrix
    dec = bp.decoder_for(csr_matrix(H))
    st = torch.cuda.current_stream(dev)

    def run():
        dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, 50, 0, 1.0, 1.0, 20.0, _lib.FLAG_FORCE_FULL,
                          hard.data_ptr(), conv.data_ptr(), iters.data_ptr(), llr.data_ptr(), st.cuda_stream)
    run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return round(B / best * 1e3)


H144 = codes.load_code("[[144, 12, 12]]").Hx
code = codes.load_code("[[288, 12, 18]]")
out["wide_864x2592"] = forced(space_time(H144, 12), 32768)
out["headline_288"] = forced(code.Hx, 125000)
dec = bp.decoder_for(code.Hx)
n = code.n
for p, osd, T in ((0.05, False, 4000000), (0.01, False, 8000000), (0.05, True, 1000000)):
    prior = np.full(n, np.log((1 - p) / p))
    fl = _lib.FLAG_OSD0 if osd else 0
    dec.mc_run(code.Lx, code.distance, p, prior, 0, 100000, flags=fl)
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        dec.mc_run(code.Lx, code.distance, p, prior, 0, T, flags=fl)
        best = min(best, time.perf_counter() - t0)
    out[f"mc_p{p}{'_osd' if osd else ''}_Mtrials_s"] = round(T / best / 1e6, 2)
# Monte-Carlo on the (8, 4) shape
st = space_time(codes.load_code("[[72, 12, 6]]").Hx, 6)
from scipy.sparse import csr_matrix
d2 = bp.decoder_for(csr_matrix(st))
Lx = np.zeros((1, st.shape[1]), np.uint8); Lx[0, :5] = 1
pr = np.full(st.shape[1], np.log(0.99 / 0.01))
d2.mc_run(Lx, 6, 0.01, pr, 0, 20000)
t0 = time.perf_counter(); d2.mc_run(Lx, 6, 0.01, pr, 0, 1000000); out["mc_wide_216x648_p0.01_Mtrials_s"] = round(1.0 / (time.perf_counter() - t0), 2)
print(json.dumps(out))
