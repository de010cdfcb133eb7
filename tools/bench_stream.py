#!/usr/bin/env python3
"""The streaming kernel (one lane per syndrome, messages in HBM: qbp_stream.hpp) on the headline
workload, device-resident, for profiling: prints one JSON line with its effective message
bandwidth.  rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this script gives the PHYSICAL HBM traffic."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=262144)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--kernel", type=int, default=_lib.KERNEL_STREAM)
ap.add_argument("--p", type=float, default=0.01)
args = ap.parse_args()
dev = torch.device("cuda", 0)
code = codes.load_code("[[288, 12, 18]]")
m, n = code.Hx.shape
E = int(code.Hx.sum())
B, p = args.batch, args.p
g = torch.Generator(device=dev); g.manual_seed(1)
err = torch.rand((B, n), generator=g, device=dev) < p
syn = (err.float() @ torch.from_numpy(code.Hx.T.astype(np.float32)).to(dev)).remainder_(2).to(torch.uint8)
prior = torch.full((n,), float(np.log((1 - p) / p)), dtype=torch.float64, device=dev)
hard = torch.empty((B, n), dtype=torch.uint8, device=dev); conv = torch.empty((B,), dtype=torch.uint8, device=dev)
iters = torch.empty((B,), dtype=torch.int32, device=dev); llr = torch.empty((B, n), dtype=torch.float64, device=dev)
dec = bp.decoder_for(code.Hx)
dec.set_option(_lib.OPT_KERNEL, args.kernel)
st = torch.cuda.current_stream(dev)


def run():
    dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, 50, 0, 1.0, 1.0, 20.0, _lib.FLAG_FORCE_FULL,
                      hard.data_ptr(), conv.data_ptr(), iters.data_ptr(), llr.data_ptr(), st.cuda_stream)


run(); torch.cuda.synchronize()
ms = []
for _ in range(args.steps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize()
    ms.append(a.elapsed_time(b))
ms = float(np.mean(ms))
alg = B * (50 * 4 * E * 8 + m + n + 8 * n + 5)
print(json.dumps({"kernel_kind": dec.info("kernel_kind"), "batch": B, "kernel_ms": ms,
                  "syndromes_per_s": B / ms * 1e3, "algorithmic_bytes_per_launch": alg,
                  "algorithmic_GBps": alg / ms / 1e6, "frac_of_8TBps": alg / ms / 1e6 / 8000}))
