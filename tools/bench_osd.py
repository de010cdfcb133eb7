#!/usr/bin/env python3
"""Device-side throughput of the OSD-0 kernel on BP failures of one code (QBP_LIB_PATH-aware)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes  # noqa: E402

dev = torch.device("cuda", 0)
out = {}
for name, p in (("[[72, 12, 6]]", 0.1), ("[[144, 12, 12]]", 0.1), ("[[288, 12, 18]]", 0.1)):
    code = codes.load_code(name)
    m, n = code.Hx.shape
    B = 131072
    g = torch.Generator(device=dev); g.manual_seed(2)
    err = torch.rand((B, n), generator=g, device=dev) < p
    syn = (err.float() @ torch.from_numpy(code.Hx.T.astype(np.float32)).to(dev)).remainder_(2).to(torch.uint8)
    prior = torch.full((n,), float(np.log((1 - p) / p)), dtype=torch.float64, device=dev)
    hard = torch.empty((B, n), dtype=torch.uint8, device=dev); conv = torch.empty((B,), dtype=torch.uint8, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev); llr = torch.empty((B, n), dtype=torch.float64, device=dev)
    sol = torch.empty((B, n), dtype=torch.uint8, device=dev)
    dec = bp.decoder_for(code.Hx)
    st = torch.cuda.current_stream(dev)
    dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, 50, 0, 1.0, 1.0, 20.0, 0, hard.data_ptr(),
                      conv.data_ptr(), iters.data_ptr(), llr.data_ptr(), st.cuda_stream)

    def run():
        dec.osd0_device(syn.data_ptr(), llr.data_ptr(), hard.data_ptr(), B, sol.data_ptr(), st.cuda_stream)
    run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    ok = bool((((sol.float() @ torch.from_numpy(code.Hx.T.astype(np.float32)).to(dev)).remainder_(2).to(torch.uint8)) == syn).all())
    out[name] = {"osd_per_s_M": round(B / best * 1e3 / 1e6, 3), "ms": round(best, 2), "all_solutions_match_syndrome": ok,
                 "checksum": int(sol.to(torch.int64).sum().item()), "bp_converged": round(float(conv.float().mean()), 3)}
print(json.dumps(out))
