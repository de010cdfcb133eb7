#!/usr/bin/env python3
"""Launch-geometry sweep of the general-H kernel (threads per workgroup x workgroups per CU), forced
50 iterations, device-resident: the reference's space-time matrices and [[288,12,18]]."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default=None)
ap.add_argument("--quick", action="store_true", help="default geometry only (A/B of library variants)")
ap.add_argument("--no-r-split", action="store_true")
ap.add_argument("--no-lds-tables", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda", 0)


def space_time(H, T):
    m = H.shape[0]
    return np.hstack([np.kron(np.eye(T, dtype=np.int64), H),
                      (np.eye(m * T, dtype=np.int64) + np.eye(m * T, k=-m, dtype=np.int64)) % 2])


H288 = codes.load_code("[[288, 12, 18]]").Hx
H144 = codes.load_code("[[144, 12, 12]]").Hx
out = {}
for name, H, B in (("288", H288, 65536), ("st864", space_time(H144, 12), 32768), ("st2592", space_time(H288, 18), 16384)):
    if args.only and args.only != name:
        continue
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
    dec.set_option(_lib.OPT_KERNEL, _lib.KERNEL_GENERAL)
    dec.set_option(_lib.OPT_GENERAL_NO_R_SPLIT, int(args.no_r_split))
    dec.set_option(_lib.OPT_GENERAL_NO_LDS_TABLES, int(args.no_lds_tables))
    st = torch.cuda.current_stream(dev)

    def run():
        dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, 50, 0, 1.0, 1.0, 20.0, _lib.FLAG_FORCE_FULL,
                          hard.data_ptr(), conv.data_ptr(), iters.data_ptr(), llr.data_ptr(), st.cuda_stream)
    res = {}
    geoms = [(0, 0)] if args.quick else [(0, 0)] + [(t, c) for t in (256, 448, 512, 640, 704, 896, 960, 1024) for c in (1, 2, 4)]
    for t, c in geoms:
        if t * c > 1024 and t:
            continue
        dec.set_option(_lib.OPT_GENERAL_THREADS, t)
        dec.set_option(_lib.OPT_BLOCKS_PER_CU, c)
        try:
            run(); torch.cuda.synchronize()
        except Exception as ex:
            res[f"{t}x{c}"] = str(ex)[:60]
            continue
        best = 1e9
        for _ in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        res[f"{dec.info('threads')}x{c or 'auto'}" + ("" if t else " (auto)")] = round(B / best * 1e3)
    out[name] = res
print(json.dumps(out))
