#!/usr/bin/env python3
"""General-H kernel (one workgroup per syndrome) against the streaming kernel (one lane per syndrome),
device-resident, forced 50 iterations: [[288,12,18]], the 864 x 2592 space-time matrix of
[[144,12,12]] and the 2592 x 7776 one of [[288,12,18]].  One JSON line (QBP_LIB_PATH-aware)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default=None, help="substring of the matrix name (profiling runs)")
ap.add_argument("--kernels", nargs="+", default=["general", "stream"])
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--batch", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda", 0)
out = {}
H288 = codes.load_code("[[288, 12, 18]]").Hx
H144 = codes.load_code("[[144, 12, 12]]").Hx


def space_time(H, T):
    m = H.shape[0]
    return np.hstack([np.kron(np.eye(T, dtype=np.int64), H),
                      (np.eye(m * T, dtype=np.int64) + np.eye(m * T, k=-m, dtype=np.int64)) % 2])


for name, H, B in (("[[288,12,18]]", H288, 65536), ("space-time 864x2592", space_time(H144, 12), 32768),
                   ("space-time 2592x7776", space_time(H288, 18), 16384)):
    if args.only and args.only not in name:
        continue
    if args.batch:
        B = args.batch
    mm, n = H.shape
    p = 0.01
    g = torch.Generator(device=dev); g.manual_seed(3)
    err = torch.rand((B, n), generator=g, device=dev) < p
    syn = (err.float() @ torch.from_numpy(H.T.astype(np.float32)).to(dev)).remainder_(2).to(torch.uint8)
    prior = torch.full((n,), float(np.log((1 - p) / p)), dtype=torch.float64, device=dev)
    hard = torch.empty((B, n), dtype=torch.uint8, device=dev); conv = torch.empty((B,), dtype=torch.uint8, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev); llr = torch.empty((B, n), dtype=torch.float64, device=dev)
    from scipy.sparse import csr_matrix
    dec = bp.decoder_for(csr_matrix(H))
    st = torch.cuda.current_stream(dev)
    out[name] = {"batch": B, "m": mm, "n": n, "edges": int(H.sum())}
    for kname, k in (("onchip", _lib.KERNEL_ON_CHIP), ("general", _lib.KERNEL_GENERAL),
                     ("stream", _lib.KERNEL_STREAM)):
        if kname not in args.kernels:
            continue
        if kname == "onchip" and (mm > 1024 or H.sum(1).max() > 8 or H.sum(0).max() > 4):
            continue
        dec.set_option(_lib.OPT_KERNEL, k)

        def run():
            dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, 50, 0, 1.0, 1.0, 20.0, _lib.FLAG_FORCE_FULL,
                              hard.data_ptr(), conv.data_ptr(), iters.data_ptr(), llr.data_ptr(), st.cuda_stream)
        run(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        out[name][kname] = {"syn_per_s": round(B / best * 1e3), "ms": round(best, 2),
                            "threads": dec.info("threads"), "grid": dec.info("grid")}
        dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_AUTO)
print(json.dumps(out))
