#!/usr/bin/env python3
"""A/B of math-evaluation variants (tools/build_variants.sh + tools/ab_run.py): forced-50 and early-exit decode
rates of [[288,12,18]], forced rate of the damped variant on [[144,12,12]]; one JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes  # noqa: E402

dev = torch.device("cuda", 0)
out = {}


def rate(name, p, B, variant, kw, flags, two_bar=False, reps=4):
    code = codes.load_code(name)
    m, n = code.Hx.shape
    g = torch.Generator(device=dev); g.manual_seed(5)
    err = torch.rand((B, n), generator=g, device=dev) < p
    syn = (err.float() @ torch.from_numpy(np.ascontiguousarray(code.Hx.T).astype(np.float32)).to(dev)).remainder_(2).to(torch.uint8)
    prior = torch.full((n,), float(np.log((1 - p) / p)), dtype=torch.float64, device=dev)
    hard = torch.empty((B, n), dtype=torch.uint8, device=dev); conv = torch.empty((B,), dtype=torch.uint8, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev); llr = torch.empty((B, n), dtype=torch.float64, device=dev)
    dec = bp.decoder_for(code.Hx)
    dec.set_option(_lib.OPT_FORCED_TWO_BARRIERS, 1 if two_bar else 0)
    st = torch.cuda.current_stream(dev)

    def run():
        dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, 50, variant, kw.get("alpha", 1.0), kw.get("damping", 1.0),
                          kw.get("clip", 20.0), flags, hard.data_ptr(), conv.data_ptr(), iters.data_ptr(), llr.data_ptr(), st.cuda_stream)
    run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    dec.set_option(_lib.OPT_FORCED_TWO_BARRIERS, 0)
    return round(B / best * 1e3 / 1e6, 3)


F = _lib.FLAG_FORCE_FULL
out["288_forced_1bar"] = rate("[[288, 12, 18]]", 0.01, 125000, 0, {}, F)
out["288_forced_2bar"] = rate("[[288, 12, 18]]", 0.01, 125000, 0, {}, F, two_bar=True)
out["288_early_p01"] = rate("[[288, 12, 18]]", 0.01, 400000, 0, {}, 0)
out["288_early_p05"] = rate("[[288, 12, 18]]", 0.05, 125000, 0, {}, 0)
out["144_damped_forced"] = rate("[[144, 12, 12]]", 0.05, 100000, 1, dict(alpha=1.0, damping=0.8, clip=20.0), F)
out["72_early_p01_10k"] = rate("[[72, 12, 6]]", 0.01, 10000, 0, {}, 0)
out["144_minsum_forced"] = rate("[[144, 12, 12]]", 0.05, 100000, 2, dict(alpha=0.8, damping=0.7, clip=25.0), F)


def mc_rate(name, p, T, flags=0, variant=0, kw=None):
    from qldpc_amd import mc
    kw = kw or {}
    code = codes.load_code(name)
    dec = bp.decoder_for(code.Hx)
    prior = torch.from_numpy(mc.prior_of(p, code.n)).to(dev)
    cnt = torch.zeros(_lib.NUM_COUNTERS, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev)

    def run():
        for a in range(0, T, 1 << 20):
            dec.mc_run_device(code.Lx, code.distance, p, prior.data_ptr(), a, min(a + (1 << 20), T), cnt.data_ptr(),
                              max_iter=50, variant=variant, flags=flags, stream=st.cuda_stream, **kw)
    run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return round(T / best * 1e3 / 1e6, 3)


out["mc288_p01"] = mc_rate("[[288, 12, 18]]", 0.01, 4 << 20)
out["mc288_p05"] = mc_rate("[[288, 12, 18]]", 0.05, 1 << 20)
out["mc288_p05_osd"] = mc_rate("[[288, 12, 18]]", 0.05, 1 << 20, flags=_lib.FLAG_OSD0)
out["mc144_minsum_p05"] = mc_rate("[[144, 12, 12]]", 0.05, 1 << 20, variant=2, kw=dict(alpha=0.8, damping=0.7, clip_llr=25.0))
print(json.dumps(out))
