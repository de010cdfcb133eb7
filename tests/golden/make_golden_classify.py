#!/usr/bin/env python3
"""Reference-executed fixture for the per-trial classification of the Monte-Carlo drivers:
tests/golden/classify.npz.

The rule lives in the loop body of /root/reference/paperResults_GPU.py:113-144 (= paperResults.py:83-100).
This generator, run in the build container only, reads that file AT GENERATION TIME, compiles the text of
those lines into a function and runs it on trials produced by the reference's own sampler and batch decoder
(decoding/beliefPropagationGPU.py) -- nothing of the reference's source is stored: the fixture holds inputs
(errors), the reference's intermediate results (syndromes, BP detections, converged flags) and the five
counters the loop body produced,

    logical_error, BPs_fault, BPs_miscorrected, incorrectable, degenerateErrors,

for three ways of filling the loop's one external call (`performOSD_enhanced(code, syndrome, llrs, detection,
order=7)` on samples BP did not converge on):
    "bp"    the call returns `detection` unchanged          (BP only: what qbp_mc_run counts without QBP_FLAG_OSD0)
    "osd0"  the reference's decoding/OSD.py performOSD       (what qbp_mc_run counts with QBP_FLAG_OSD0)
    "osdw"  the reference's performOSD_enhanced itself (order 7), on the smaller cases: equal to "osd0" on
            every one of them (it returns its OSD-0 solution whenever that reproduces the syndrome).

    MPLBACKEND=Agg python tests/golden/make_golden_classify.py
"""
import os
import sys
import textwrap
import time

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")
from decoding.beliefPropagationGPU import generate_errors_and_syndromes_batch, performBeliefPropagationBatch  # noqa: E402
from decoding.OSD import performOSD  # noqa: E402
from decoding.OSD_enhanced import performOSD_enhanced  # noqa: E402

FIRST, LAST = 113, 144          # 1-based, inclusive: the per-sample loop of the batch driver


def compile_loop_body():
    lines = open(os.path.join(REF, "paperResults_GPU.py")).read().splitlines()[FIRST - 1:LAST]
    assert lines[0].strip().startswith("for i in range(current_batch_size):"), lines[0]
    assert lines[-1].strip() == "incorrectable += 1", lines[-1]
    body = textwrap.indent(textwrap.dedent("\n".join(lines)), "    ")
    src = ("def loop(current_batch_size, errors, syndromes, detections, llrs_batch, converged, code, Lx, distance,\n"
           "         performOSD_enhanced):\n"
           "    logical_error = 0; BPs_fault = 0; BPs_miscorrected = 0; incorrectable = 0; degenerateErrors = 0\n"
           + body + "\n"
           "    return logical_error, BPs_fault, BPs_miscorrected, incorrectable, degenerateErrors\n")
    ns = {"np": np}
    exec(compile(src, "<paperResults_GPU.py:113-144>", "exec"), ns)
    return ns["loop"]


def main():
    loop = compile_loop_body()
    out = {}
    cases = [  # name, code file, p, trials, double draw, maxIter, chunk, also order-7 search
        ("72_p0.05", "[[72, 12, 6]]", 0.05, 1500, False, 50, 500, True),
        ("72_p0.10", "[[72, 12, 6]]", 0.10, 1000, False, 50, 500, True),
        ("72_p0.03_xor", "[[72, 12, 6]]", 0.03, 1500, True, 150, 500, False),      # the driver's own noise model / limit
        ("288_p0.06", "[[288, 12, 18]]", 0.06, 800, False, 50, 100, False),
        ("288_p0.08", "[[288, 12, 18]]", 0.08, 600, False, 50, 100, False),
    ]
    rng = np.random.default_rng(20261005)
    for name, fname, p, trials, xor, max_iter, chunk, with_w in cases:
        d = np.load(os.path.join(REF, "codes", f"{fname}.npz"))
        code, Lx, distance = d["Hx"], d["Lx"], int(d["distance"])
        n = code.shape[1]
        prior = np.array([np.log((1 - p) / p)] * n)
        E, S, D, C, DO = [], [], [], [], []
        cnt = {k: np.zeros(5, np.int64) for k in ("bp", "osd0", "osdw")}
        t0 = time.time()
        for lo in range(0, trials, chunk):
            b = min(chunk, trials - lo)
            errors, syndromes = generate_errors_and_syndromes_batch(code, p, b, rng)
            if xor:                                                   # paperResults_GPU.py:96-105
                e2, s2 = generate_errors_and_syndromes_batch(code, p, b, rng)
                errors, syndromes = (errors + e2) % 2, (syndromes + s2) % 2
            det, conv, llrs = performBeliefPropagationBatch(code, syndromes, prior, maxIter=max_iter)
            osd_out = det.copy()

            def osd0_call(code_, syndrome, llrs_, detection, order=7):
                return performOSD(code_, syndrome, llrs_, detection)
            fills = {"bp": lambda code_, syndrome, llrs_, detection, order=7: detection, "osd0": osd0_call}
            if with_w:
                fills["osdw"] = performOSD_enhanced
            for k, fill in fills.items():
                cnt[k] += np.array(loop(b, errors, syndromes, det, llrs, conv, code, Lx, distance, fill), np.int64)
            for i in np.flatnonzero(~conv):
                osd_out[i] = performOSD(code, syndromes[i], llrs[i], det[i])
            E.append(errors); S.append(syndromes); D.append(det); C.append(conv); DO.append(osd_out)
        if with_w:
            assert np.array_equal(cnt["osdw"], cnt["osd0"]), (name, cnt)
        E, S, D, C, DO = (np.concatenate(x) for x in (E, S, D, C, DO))
        out[f"{name}/errors"] = np.packbits(E.astype(np.uint8), axis=1)
        out[f"{name}/syndromes"] = np.packbits(S.astype(np.uint8), axis=1)
        out[f"{name}/detections_bp"] = np.packbits(D.astype(np.uint8), axis=1)
        out[f"{name}/detections_osd0"] = np.packbits(DO.astype(np.uint8), axis=1)
        out[f"{name}/converged"] = C.astype(np.uint8)
        out[f"{name}/counters_bp"] = cnt["bp"]
        out[f"{name}/counters_osd0"] = cnt["osd0"]
        out[f"{name}/meta"] = np.array([p, trials, int(xor), max_iter, distance, int(with_w)], np.float64)
        print(f"{name}: {trials} trials, {int((~C).sum())} not converged; counters (logical_error, BPs_fault, "
              f"BPs_miscorrected, incorrectable, degenerateErrors): BP only {cnt['bp'].tolist()}, "
              f"with OSD-0 {cnt['osd0'].tolist()}" + (", order-7 search identical" if with_w else "")
              + f"  [{time.time() - t0:.0f} s]", flush=True)
    out["names"] = np.array([c[0] for c in cases])
    out["code_of"] = np.array([c[1] for c in cases])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "classify.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
