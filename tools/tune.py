#!/usr/bin/env python3
"""Sweep launch geometry of the fused kernel on the GPU box: slots per workgroup x workgroups per
CU, [[288,12,18]] forced 50 iterations.  Prints one line per config."""
import argparse
import itertools
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--code", default="[[288, 12, 18]]")
    ap.add_argument("--batch", type=int, default=40000)
    ap.add_argument("--p", type=float, default=0.01)
    ap.add_argument("--slots", type=int, nargs="+", default=[1, 2, 3, 4, 5, 6, 7])
    ap.add_argument("--blocks", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--early-exit", action="store_true")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    code = codes.load_code(args.code)
    m, n = code.Hx.shape
    B = args.batch
    g = torch.Generator(device=dev); g.manual_seed(1)
    err = torch.rand((B, n), generator=g, device=dev) < args.p
    syn = (err.float() @ torch.from_numpy(code.Hx.T.astype(np.float32)).to(dev)).remainder_(2).to(torch.uint8)
    prior = torch.full((n,), float(np.log((1 - args.p) / args.p)), dtype=torch.float64, device=dev)
    hard = torch.empty((B, n), dtype=torch.uint8, device=dev)
    conv = torch.empty((B,), dtype=torch.uint8, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev)
    llr = torch.empty((B, n), dtype=torch.float64, device=dev)
    dec = bp.decoder_for(code.Hx)
    stream = torch.cuda.current_stream(dev)
    flags = 0 if args.early_exit else _lib.FLAG_FORCE_FULL
    results = []
    for S, blocks in itertools.product(args.slots, args.blocks):
        if S * m > 1024:
            continue
        threads = ((S * m + 63) // 64) * 64
        dec.set_option(_lib.OPT_SLOTS_PER_BLOCK, S)
        dec.set_option(_lib.OPT_BLOCKS_PER_CU, blocks)

        def run():
            dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, 50, 0, 1.0, 1.0, 20.0, flags,
                              hard.data_ptr(), conv.data_ptr(), iters.data_ptr(), llr.data_ptr(),
                              stream.cuda_stream)
        run(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        r = dict(S=S, blocks_per_cu=blocks, threads=dec.info("threads"),
                 grid=dec.info("grid"), lds=dec.info("lds_bytes"), ms=best, syn_per_s=B / best * 1e3)
        results.append(r)
        print(json.dumps(r), flush=True)
    best = max(results, key=lambda r: r["syn_per_s"])
    print("BEST", json.dumps(best))


if __name__ == "__main__":
    main()
