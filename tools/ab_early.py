#!/usr/bin/env python3
"""Early-exit (reference semantics) throughput of the library named by QBP_LIB_PATH: device-resident
decode of [[288,12,18]] at several error rates, and the Monte-Carlo loop.  One line of JSON."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes  # noqa: E402

dev = torch.device("cuda", 0)
code = codes.load_code("[[288, 12, 18]]")
m, n = code.Hx.shape
dec = bp.decoder_for(code.Hx)
st = torch.cuda.current_stream(dev)
out = {}
B = 400000
hard = torch.empty((B, n), dtype=torch.uint8, device=dev); conv = torch.empty((B,), dtype=torch.uint8, device=dev)
iters = torch.empty((B,), dtype=torch.int32, device=dev); llr = torch.empty((B, n), dtype=torch.float64, device=dev)
HxT = torch.from_numpy(code.Hx.T.astype(np.float32)).to(dev)
for p in (0.01, 0.03, 0.05):
    g = torch.Generator(device=dev); g.manual_seed(1)
    err = torch.rand((B, n), generator=g, device=dev) < p
    syn = (err.float() @ HxT).remainder_(2).to(torch.uint8)
    prior = torch.full((n,), float(np.log((1 - p) / p)), dtype=torch.float64, device=dev)

    def run():
        dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, 50, 0, 1.0, 1.0, 20.0, 0, hard.data_ptr(),
                          conv.data_ptr(), iters.data_ptr(), llr.data_ptr(), st.cuda_stream)
    run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    out[f"decode p={p}"] = round(B / best * 1e3 / 1e6, 2)
    out[f"iters p={p}"] = round(float(iters.float().mean()) + 1, 2)
for p, osd in ((0.01, False), (0.05, False), (0.05, True)):
    T = 4000000 if not osd else 1000000
    prior = np.full(n, np.log((1 - p) / p))
    # (warm-up at full size: the per-trial records of the OSD pass are allocated by the first call that needs them)
    dec.mc_run(code.Lx, code.distance, p, prior, 0, T, flags=_lib.FLAG_OSD0 if osd else 0)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    dec.mc_run(code.Lx, code.distance, p, prior, 0, T, flags=_lib.FLAG_OSD0 if osd else 0)
    dt = time.perf_counter() - t0
    out[f"mc p={p}{' osd' if osd else ''}"] = round(T / dt / 1e6, 2)
# the reference driver's batch (paperResults_GPU.py:44,108): 5 000 syndromes, two draws at p = 0.05, maxIter 150
B2 = 5000
g = torch.Generator(device=dev); g.manual_seed(2)
err = (torch.rand((B2, n), generator=g, device=dev) < 0.05) ^ (torch.rand((B2, n), generator=g, device=dev) < 0.05)
syn = (err.float() @ HxT).remainder_(2).to(torch.uint8)
prior = torch.full((n,), float(np.log(0.95 / 0.05)), dtype=torch.float64, device=dev)
for _ in range(2):
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dec.decode_device(syn.data_ptr(), prior.data_ptr(), B2, 150, 0, 1.0, 1.0, 20.0, 0, hard.data_ptr(),
                          conv.data_ptr(), iters.data_ptr(), llr.data_ptr(), st.cuda_stream)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
out["driver batch 5000 ms"] = round(best, 3)
print(json.dumps(out))
