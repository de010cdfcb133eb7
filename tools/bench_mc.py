#!/usr/bin/env python3
"""Device-resident Monte-Carlo loop (qbp_mc_run_device: sample + decode + classify [+ OSD-0]) of [[288,12,18]]:
trials/s at p = 0.01 and 0.05, BP(50), with and without OSD-0 (QBP_LIB_PATH-aware; also the target of the PMC
passes of tools/profile_r03.sh)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qldpc_amd import _lib, bp, codes, mc  # noqa: E402

dev = torch.device("cuda", 0)
code = codes.load_code("[[288, 12, 18]]")
dec = bp.decoder_for(code.Hx)
st = torch.cuda.current_stream(dev)
out = {}
for p, T, flags, tag in ((0.01, 4 << 20, 0, "p0.01"), (0.05, 1 << 20, 0, "p0.05"), (0.05, 1 << 20, _lib.FLAG_OSD0, "p0.05+osd0")):
    prior = torch.from_numpy(mc.prior_of(p, code.n)).to(dev)
    cnt = torch.zeros(_lib.NUM_COUNTERS, dtype=torch.int64, device=dev)

    def run():
        for a in range(0, T, 1 << 20):
            dec.mc_run_device(code.Lx, code.distance, p, prior.data_ptr(), a, min(a + (1 << 20), T), cnt.data_ptr(),
                              max_iter=50, flags=flags, stream=st.cuda_stream)
    run(); torch.cuda.synchronize()
    cnt.zero_()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    c = (cnt // 3).cpu().numpy()
    out[tag] = {"trials_per_s_M": round(T / best * 1e3 / 1e6, 2), "ms": round(best, 2), "ler": float(c[1] / c[0]),
                "not_converged": float(c[6] / c[0]), "mean_iterations": float(c[7] / c[0] + 1)}
print(json.dumps(out))
