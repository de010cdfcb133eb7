#!/usr/bin/env python3
"""GPU: posterior LLRs of every device kernel against the CPU oracle, BIT FOR BIT (round 3: both use
numpy's own tanh / arctanh kernels).  Prints one line per configuration; exit code 1 on any mismatch.

    python tools/check_exact.py [--n 2000]
"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import oracle  # noqa: E402
from qldpc_amd import _lib, bp, codes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    a = ap.parse_args()
    bad_total = 0
    rng = np.random.default_rng(77)
    for tag in ("72", "144", "288"):
        code = codes.load_code(tag)
        H = np.ascontiguousarray(code.Hx)
        for p in (0.03, 0.08):
            e = (rng.random((a.n, code.n)) < p).astype(np.uint8)
            s = (e @ H.T % 2).astype(np.uint8)
            prior = np.full(code.n, np.log((1 - p) / p))
            for variant, kw in ((0, {}), (1, dict(alpha=1.0, damping=0.8, clip_llr=20.0)),
                                (2, dict(alpha=0.8, damping=0.7, clip_llr=25.0))):
                t0 = time.time()
                oh, oc, oi, ol = oracle.decode_batch(H, s, prior, 50, variant, threads=8, **kw)
                t_or = time.time() - t0
                for kname, opt in (("fused", None), ("generic", (_lib.OPT_KERNEL, 2)), ("stream", (_lib.OPT_KERNEL, 3))):
                    dec = _lib.Decoder(*bp.csr_from_H(H), bp.DEVICE)
                    if opt:
                        dec.set_option(*opt)
                    for flags in (0, _lib.FLAG_FORCE_FULL):
                        h, c, it, llr = dec.decode(s, prior, 50, variant, kw.get("alpha", 1.0), kw.get("damping", 1.0),
                                                   kw.get("clip_llr", 20.0), flags)
                        rows = (llr.view(np.uint64) != ol.view(np.uint64)).any(1)
                        nb = int(rows.sum()) + int((h != oh).any(1).sum()) + int((c != oc).sum()) + int((it != oi).sum())
                        bad_total += nb
                        rel = np.abs(llr - ol) / np.maximum(np.abs(ol), 1e-300)
                        print(f"[[{tag}]] p={p} variant {variant} {kname:7s} flags {flags}: kernel kind "
                              f"{dec.info('kernel_kind')}, LLR rows differing {int(rows.sum())}/{a.n} "
                              f"(max rel {rel.max():.1e}), hard/conv/iter mismatches "
                              f"{int((h != oh).any(1).sum())}/{int((c != oc).sum())}/{int((it != oi).sum())}"
                              f"  [oracle {t_or:.1f}s]", flush=True)
                    dec.close()
    print("TOTAL mismatches", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
