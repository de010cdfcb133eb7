#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (parity-test cases, not the bench line):
config 2 [[72,12,6]] p=0.01 sum-product 10k batch; config 3 [[144,12,12]] min-sum alpha 0.8 damping
0.7 clip 25, 100k batch; config 4 shape on one GPU.  Device-resident inputs, HIP-event timing."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes  # noqa: E402

dev = torch.device("cuda", 0)
CASES = [("config2", "[[72, 12, 6]]", 0.01, 10_000, _lib.SUM_PRODUCT, dict(alpha=1.0, damping=1.0, clip=20.0)),
         ("config2-large-batch", "[[72, 12, 6]]", 0.01, 1_000_000, _lib.SUM_PRODUCT, dict(alpha=1.0, damping=1.0, clip=20.0)),
         ("config3 p=0.05", "[[144, 12, 12]]", 0.05, 100_000, _lib.MIN_SUM, dict(alpha=0.8, damping=0.7, clip=25.0)),
         ("config3 p=0.02", "[[144, 12, 12]]", 0.02, 100_000, _lib.MIN_SUM, dict(alpha=0.8, damping=0.7, clip=25.0)),
         ("damped-SP 144", "[[144, 12, 12]]", 0.05, 100_000, _lib.DAMPED_SP, dict(alpha=1.0, damping=0.8, clip=20.0)),
         ("config4 shard", "[[288, 12, 18]]", 0.01, 125_000, _lib.SUM_PRODUCT, dict(alpha=1.0, damping=1.0, clip=20.0))]
for tag, name, p, B, variant, kw in CASES:
    code = codes.load_code(name)
    m, n = code.Hx.shape
    g = torch.Generator(device=dev); g.manual_seed(5)
    err = torch.rand((B, n), generator=g, device=dev) < p
    syn = (err.float() @ torch.from_numpy(code.Hx.T.astype(np.float32)).to(dev)).remainder_(2).to(torch.uint8)
    prior = torch.full((n,), float(np.log((1 - p) / p)), dtype=torch.float64, device=dev)
    hard = torch.empty((B, n), dtype=torch.uint8, device=dev)
    conv = torch.empty((B,), dtype=torch.uint8, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    llr = torch.empty((B, n), dtype=torch.float64, device=dev)
    dec = bp.decoder_for(code.Hx)
    st = torch.cuda.current_stream(dev)
    out = {"case": tag, "code": name, "p": p, "batch": B, "variant": int(variant), **kw}
    for mode, flags in (("early_exit", 0), ("forced_50", _lib.FLAG_FORCE_FULL)):
        def run():
            dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, 50, variant, kw["alpha"], kw["damping"],
                              kw["clip"], flags, hard.data_ptr(), conv.data_ptr(), iters.data_ptr(),
                              llr.data_ptr(), st.cuda_stream)
        run(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        out[mode + "_syn_per_s"] = B / best * 1e3
        out[mode + "_ms"] = best
    out["mean_iterations"] = float(iters.double().mean().item()) + 1.0
    out["converged"] = float(conv.double().mean().item())
    print(json.dumps(out), flush=True)
