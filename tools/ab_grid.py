#!/usr/bin/env python3
"""Queued workgroups per CU of the on-chip kernel (QBP_OPT_BLOCKS_PER_CU: 1 = resident only, everything
beyond a slot's first syndrome comes from the work counter; 2 = default) on early-exit workloads."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes  # noqa: E402

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)


def case(name, B, p, max_iter, draws=1):
    code = codes.load_code(name)
    m, n = code.Hx.shape
    dec = bp.decoder_for(code.Hx)
    g = torch.Generator(device=dev); g.manual_seed(1)
    err = torch.rand((B, n), generator=g, device=dev) < p
    if draws == 2:
        err ^= torch.rand((B, n), generator=g, device=dev) < p
    HxT = torch.from_numpy(code.Hx.T.astype(np.float32)).to(dev)
    syn = (err.float() @ HxT).remainder_(2).to(torch.uint8)
    prior = torch.full((n,), float(np.log((1 - p) / p)), dtype=torch.float64, device=dev)
    hard = torch.empty((B, n), dtype=torch.uint8, device=dev); conv = torch.empty((B,), dtype=torch.uint8, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev); llr = torch.empty((B, n), dtype=torch.float64, device=dev)
    res = {}
    for per_cu in (1, 2, 3, 1, 2, 3):
        dec.set_option(_lib.OPT_BLOCKS_PER_CU, per_cu)
        best = 1e9
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, max_iter, 0, 1.0, 1.0, 20.0, 0, hard.data_ptr(),
                              conv.data_ptr(), iters.data_ptr(), llr.data_ptr(), st.cuda_stream)
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        res.setdefault(per_cu, []).append(round(best, 4))
    dec.set_option(_lib.OPT_BLOCKS_PER_CU, 0)
    print(json.dumps({"case": f"{name} B={B} p={p} maxIter={max_iter} draws={draws}",
                      "mean_iters": round(float(iters.float().mean()) + 1, 2), "ms_by_blocks_per_cu": res}), flush=True)


case("[[288, 12, 18]]", 400000, 0.01, 50)
case("[[288, 12, 18]]", 125000, 0.01, 50)
case("[[288, 12, 18]]", 400000, 0.05, 50)
case("[[288, 12, 18]]", 5000, 0.05, 150, 2)
case("[[288, 12, 18]]", 20000, 0.05, 150, 2)
case("[[144, 12, 12]]", 100000, 0.02, 50)
case("[[72, 12, 6]]", 10000, 0.01, 50)
case("[[72, 12, 6]]", 1000000, 0.01, 50)
