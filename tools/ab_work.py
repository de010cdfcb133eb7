#!/usr/bin/env python3
"""Early-exit kernel times (ms, best of 7) of the library named by QBP_LIB_PATH on batches of several
sizes and difficulties, default launch geometry: the work-distribution A/B (tools/ab_run.py tools/ab_work.py)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes  # noqa: E402

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
out = {}


def case(tag, name, B, p, max_iter, draws=1, variant=0, alpha=1.0, damping=1.0, clip=20.0):
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
    best = 1e9
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, max_iter, variant, alpha, damping, clip, 0, hard.data_ptr(),
                          conv.data_ptr(), iters.data_ptr(), llr.data_ptr(), st.cuda_stream)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    out[tag] = round(best, 4)


case("288 easy 400k", "[[288, 12, 18]]", 400000, 0.01, 50)
case("288 easy 125k", "[[288, 12, 18]]", 125000, 0.01, 50)
case("288 easy 40k", "[[288, 12, 18]]", 40000, 0.01, 50)
case("288 easy 20k", "[[288, 12, 18]]", 20000, 0.01, 50)
case("288 p.03 125k", "[[288, 12, 18]]", 125000, 0.03, 50)
case("288 p.05 400k", "[[288, 12, 18]]", 400000, 0.05, 50)
case("288 p.05 20k", "[[288, 12, 18]]", 20000, 0.05, 50)
case("288 hard 5k", "[[288, 12, 18]]", 5000, 0.05, 150, 2)
case("288 hard 20k", "[[288, 12, 18]]", 20000, 0.05, 150, 2)
case("288 hard 100k", "[[288, 12, 18]]", 100000, 0.05, 150, 2)
case("144ms p.02 100k", "[[144, 12, 12]]", 100000, 0.02, 50, 1, 2, 0.8, 0.7, 25.0)
case("144ms p.05 100k", "[[144, 12, 12]]", 100000, 0.05, 50, 1, 2, 0.8, 0.7, 25.0)
case("72 easy 10k", "[[72, 12, 6]]", 10000, 0.01, 50)
case("72 easy 100k", "[[72, 12, 6]]", 100000, 0.01, 50)
case("72 easy 1M", "[[72, 12, 6]]", 1000000, 0.01, 50)
print(json.dumps(out))
