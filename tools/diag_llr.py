#!/usr/bin/env python3
"""Where does the device LLR differ from the oracle's by more than 1e-5 relative?  (GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from qldpc_amd import bp, codes  # noqa: E402

for name, p, B in (("[[144, 12, 12]]", 0.05, 4000), ("[[288, 12, 18]]", 0.05, 3000),
                   ("[[288, 12, 18]]", 0.08, 2000)):
    code = codes.load_code(name)
    rng = np.random.default_rng([code.n, int(p * 1e6)])
    syn = ((rng.random((B, code.n)) < p).astype(np.uint8) @ code.Hx.T % 2).astype(np.uint8)
    prior = np.full(code.n, np.log((1 - p) / p))
    hard, conv, iters, llr = bp.decoder_for(code.Hx).decode(syn, prior, 50)
    o_hard, o_conv, o_iters, o_llr = oracle.decode_batch(code.Hx, syn, prior, 50)
    print(name, p, "conv equal", np.array_equal(conv, o_conv), "iters equal",
          np.array_equal(iters, o_iters), "hard equal rows", int((hard == o_hard).all(1).sum()), "/", B)
    rel = np.abs(llr - o_llr) / np.maximum(np.abs(o_llr), 1e-300)
    relrow = rel.max(1)
    for lo, hi in ((0, 1), (2, 5), (6, 10), (11, 20), (21, 35), (36, 49)):
        sel = conv & (iters >= lo) & (iters <= hi)
        if sel.any():
            print(f"  converged at iter {lo:2d}-{hi:2d}: {int(sel.sum()):5d} syndromes, max rel {relrow[sel].max():.2e}, "
                  f"frac > 1e-5: {np.mean(relrow[sel] > 1e-5):.4f}, max abs {np.abs(llr - o_llr)[sel].max():.2e}")
    bad = np.argwhere((rel > 1e-5) & conv[:, None])
    for b, v in bad[:8]:
        print(f"  syndrome {b} iter {iters[b]} var {v}: hip {llr[b, v]!r} oracle {o_llr[b, v]!r} rel {rel[b, v]:.2e}")
    nc = ~conv
    if nc.any():
        print(f"  non-converged: {int(nc.sum())}, hard rows equal {int((hard[nc] == o_hard[nc]).all(1).sum())}, "
              f"max abs LLR diff {np.abs(llr - o_llr)[nc].max():.2e}")
