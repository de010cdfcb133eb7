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

# the reference driver's batch call (paperResults_GPU.py:108: 5000 syndromes, maxIter 150), host
# arrays in and out: where the time of one call goes
code = codes.load_code("[[288, 12, 18]]")
H = code.Hx
n = code.n
p = 0.05
rng = np.random.default_rng(0)
e = ((rng.random((5000, n)) < p) ^ (rng.random((5000, n)) < p)).astype(np.int64)
syn = (e @ H.T % 2)
prior = np.full(n, np.log((1 - p) / p))
dec = bp.decoder_for(H)
syn8 = syn.astype(np.uint8)
dec.decode(syn8, prior, 150)                      # buffers of this size exist from here on
t0 = time.perf_counter()
for _ in range(5):
    dec.decode(syn8, prior, 150)
t_c = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for _ in range(5):
    dec.decode(syn8, prior, 150, want_llr=False)
t_nollr = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for _ in range(5):
    bp.performBeliefPropagationBatch(H, syn, prior, maxIter=150)
t_py = (time.perf_counter() - t0) / 5
import torch  # noqa: E402
dev = torch.device("cuda", 0)
d_syn = torch.from_numpy(syn8).to(dev); d_pr = torch.from_numpy(prior).to(dev)
d_hard = torch.empty((5000, n), dtype=torch.uint8, device=dev); d_conv = torch.empty(5000, dtype=torch.uint8, device=dev)
d_it = torch.empty(5000, dtype=torch.int32, device=dev); d_llr = torch.empty((5000, n), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream(dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dec.decode_device(d_syn.data_ptr(), d_pr.data_ptr(), 5000, 150, 0, 1.0, 1.0, 20.0, 0, d_hard.data_ptr(),
                      d_conv.data_ptr(), d_it.data_ptr(), d_llr.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
t_k = (time.perf_counter() - t0) / 5
print(f"[[288, 12, 18]] batch of 5000 (two draws at p = 0.05, maxIter 150): kernel alone {t_k * 1e3:.2f} ms; "
      f"qbp_decode_batch (host arrays, 13 MB back) {t_c * 1e3:.2f} ms, without the LLR array {t_nollr * 1e3:.2f} ms; "
      f"performBeliefPropagationBatch (Python mirror: casts, checks, int8 / bool views) {t_py * 1e3:.2f} ms")
