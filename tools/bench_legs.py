#!/usr/bin/env python3
"""The other BASELINE.json configurations on one GPU -- the workloads of bench.py's `other_configs` legs.

Two users, one definition (so that the hardware counters and the timings belong to the same launches):

* ``bench.py`` imports :func:`legs` and times every leg live (HIP events on the launch stream);
* ``tools/profile_r03.sh`` runs ``python3 tools/bench_legs.py --once`` under ``rocprofv3 --pmc``: every leg then
  launches its kernel exactly once, in the order of :data:`ORDER`, and ``tools/pmc_collect.py`` maps the k-th
  dispatch of a qbp kernel to the k-th entry -- the committed summary (profiles/r03_pmc_legs.json) gives bench.py
  the instruction counts it prices the live timings with.

Inputs are drawn on the device from fixed seeds: the same launches every time.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qldpc_amd import _lib, bp, codes, mc  # noqa: E402

MAX_ITER = 50
# (leg name, qbp kernels it launches, in order; osd0_288: the decode, the OSD sweep, its redo pass for inconsistent
# syndromes)
ORDER = [("config2_early_exit", 1), ("config2_forced_50", 1), ("config3_early_exit", 1), ("config3_forced_50", 1),
         ("mc288_p0.01", 1), ("mc288_p0.05", 1), ("mc288_p0.05_osd0", 2), ("osd0_288", 3)]


def _syndromes(dev, code, p, B, seed):
    m, n = code.Hx.shape
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    err = torch.rand((B, n), generator=g, device=dev) < p
    Ht = torch.from_numpy(np.ascontiguousarray(code.Hx.T).astype(np.float32)).to(dev)
    return (err.float() @ Ht).remainder_(2).to(torch.uint8).contiguous(), Ht


def legs(device):
    """Yield (name, info dict, launch callable, after callable -> dict of results read back)."""
    dev = torch.device("cuda", device)
    st = torch.cuda.current_stream(dev)
    for tag, name, p, B, variant, kw in (
            ("config2", "[[72, 12, 6]]", 0.01, 10_000, _lib.SUM_PRODUCT, dict(alpha=1.0, damping=1.0, clip=20.0)),
            ("config3", "[[144, 12, 12]]", 0.05, 100_000, _lib.MIN_SUM, dict(alpha=0.8, damping=0.7, clip=25.0))):
        code = codes.load_code(name)
        m, n = code.Hx.shape
        syn, _ = _syndromes(dev, code, p, B, 5)
        prior = torch.full((n,), float(np.log((1 - p) / p)), dtype=torch.float64, device=dev)
        hard = torch.empty((B, n), dtype=torch.uint8, device=dev); conv = torch.empty((B,), dtype=torch.uint8, device=dev)
        iters = torch.empty((B,), dtype=torch.int32, device=dev); llr = torch.empty((B, n), dtype=torch.float64, device=dev)
        dec = bp.decoder_for(code.Hx, device=device)
        for mode, flags in (("early_exit", 0), ("forced_50", _lib.FLAG_FORCE_FULL)):
            def run(flags=flags, dec=dec, syn=syn, prior=prior, hard=hard, conv=conv, iters=iters, llr=llr, B=B,
                    variant=variant, kw=kw):
                dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, MAX_ITER, variant, kw["alpha"], kw["damping"],
                                  kw["clip"], flags, hard.data_ptr(), conv.data_ptr(), iters.data_ptr(), llr.data_ptr(),
                                  st.cuda_stream)

            def after(flags=flags, iters=iters, conv=conv, B=B, dec=dec, m=m):
                it_total = B * MAX_ITER if flags else int(iters.sum(dtype=torch.int64).item()) + B
                return {"units": B, "unit": "syndromes/s", "iterations_total": it_total,
                        "converged_fraction": float(conv.float().mean().item()), "threads": dec.info("threads"),
                        "grid": dec.info("grid"), "m": m}
            info = {"workload": f"{name} p={p} B={B} variant={int(variant)} " + " ".join(f"{k}={v}" for k, v in kw.items()),
                    "E": int(code.Hx.sum()), "m": m, "n": n}
            yield f"{tag}_{mode}", info, run, after
    code = codes.load_code("[[288, 12, 18]]")
    m, n = code.Hx.shape
    dec = bp.decoder_for(code.Hx, device=device)
    for p, T, flags, tag in ((0.01, 1 << 20, 0, "mc288_p0.01"), (0.05, 1 << 20, 0, "mc288_p0.05"),
                             (0.05, 1 << 20, _lib.FLAG_OSD0, "mc288_p0.05_osd0")):
        prior = torch.from_numpy(mc.prior_of(p, n)).to(dev)
        cnt = torch.zeros(_lib.NUM_COUNTERS, dtype=torch.int64, device=dev)

        def run(p=p, T=T, flags=flags, prior=prior, cnt=cnt):
            cnt.zero_()
            dec.mc_run_device(code.Lx, code.distance, p, prior.data_ptr(), 0, T, cnt.data_ptr(), max_iter=MAX_ITER,
                              flags=flags, stream=st.cuda_stream)

        def after(cnt=cnt, T=T):
            c = cnt.cpu().numpy()
            return {"units": T, "unit": "trials/s", "iterations_total": int(c[7] + c[0]), "ler": float(c[1] / c[0]),
                    "not_converged": float(c[6] / c[0]), "threads": dec.info("threads"), "grid": dec.info("grid"), "m": m}
        yield tag, {"workload": f"[[288,12,18]] qbp_mc_run_device p={p} T={T} flags={flags}", "E": int(code.Hx.sum()),
                    "m": m, "n": n}, run, after
    # OSD-0 alone: the BP failures of a p = 0.1 batch (first kernel: the decode that produces them)
    B, p = 131072, 0.1
    syn, Ht = _syndromes(dev, code, p, B, 2)
    prior = torch.full((n,), float(np.log((1 - p) / p)), dtype=torch.float64, device=dev)
    hard = torch.empty((B, n), dtype=torch.uint8, device=dev); conv = torch.empty((B,), dtype=torch.uint8, device=dev)
    iters = torch.empty((B,), dtype=torch.int32, device=dev); llr = torch.empty((B, n), dtype=torch.float64, device=dev)
    sol = torch.empty((B, n), dtype=torch.uint8, device=dev)
    state = {"decoded": False}

    def run():
        if not state["decoded"]:
            dec.decode_device(syn.data_ptr(), prior.data_ptr(), B, MAX_ITER, 0, 1.0, 1.0, 20.0, 0, hard.data_ptr(),
                              conv.data_ptr(), iters.data_ptr(), llr.data_ptr(), st.cuda_stream)
            state["decoded"] = True
        dec.osd0_device(syn.data_ptr(), llr.data_ptr(), hard.data_ptr(), B, sol.data_ptr(), st.cuda_stream)

    def after():
        ok = bool((((sol.float() @ Ht).remainder_(2).to(torch.uint8)) == syn).all())
        return {"units": B, "unit": "OSD-0 solutions/s", "all_solutions_reproduce_their_syndrome": ok,
                "bp_converged": float(conv.float().mean().item())}
    yield "osd0_288", {"workload": f"[[288,12,18]] qbp_osd0_batch_device on the outputs of a p={p} batch of {B}"}, run, after


def main():
    once = "--once" in sys.argv
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    out = {}
    for name, info, run, after in legs(0):
        if once:
            run()
            torch.cuda.synchronize(dev)
            out[name] = after()
            continue
        run(); torch.cuda.synchronize(dev)
        ms = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); run(); b.record(st); torch.cuda.synchronize(dev)
            ms.append(a.elapsed_time(b))
        r = after()
        r["ms"] = float(np.median(ms))
        r["value"] = r["units"] / r["ms"] * 1e3
        out[name] = r
    print(json.dumps(out))


if __name__ == "__main__":
    main()
