#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by RUNNING THE REAL REFERENCE.

Build-container only (``/root/reference`` never travels to the GPU box); the outputs
(small ``.npz`` files of inputs and expected outputs) are committed.

    MPLBACKEND=Agg python tests/golden/make_golden.py

Reference entry points exercised (paths relative to /root/reference):
  fast3   decoding/beliefPropagation.py:88   performBeliefPropagationFast  -> 3-tuple
  loop3   decoding/beliefPropagation.py:6    performBeliefPropagation      -> 3-tuple
  fast4   rework/decoding.py:77              performBeliefPropagationFast  -> 4-tuple (+iteration)
  minsum  rework/decoding.py:5               performMinSum_Symmetric       -> 4-tuple
  sym     rework/decoding.py:131             performBeliefPropagation_Symmetric -> 4-tuple
  batch   decoding/beliefPropagationGPU.py:81 performBeliefPropagationBatch (NumPy fallback)
Inputs come from decoding/beliefPropagationGPU.py:181 generate_errors_and_syndromes_batch
with ``np.random.default_rng(20260128)``.
"""
import contextlib
import importlib.util
import io
import json
import os
import sys

import numpy as np

REF = os.environ.get("QLDPC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, REF)

with contextlib.redirect_stdout(io.StringIO()):
    from decoding.beliefPropagation import (performBeliefPropagation,        # noqa: E402
                                            performBeliefPropagationFast)
    from decoding.beliefPropagationGPU import (generate_errors_and_syndromes_batch,  # noqa: E402
                                               performBeliefPropagationBatch)
_spec = importlib.util.spec_from_file_location("ref_rework_decoding",
                                               os.path.join(REF, "rework", "decoding.py"))
rework = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(rework)

SEED = 20260128


def prior_of(p, n):
    return np.array([np.log((1 - p) / p)] * n)     # main.py:18, paperResults.py:49


def run_single(fn_name, H, syndromes, prior, max_iter, **kw):
    B, n = syndromes.shape[0], H.shape[1]
    hard = np.zeros((B, n), np.uint8)
    conv = np.zeros(B, bool)
    iters = np.full(B, -1, np.int32)
    llr = np.zeros((B, n))
    for i in range(B):
        with contextlib.redirect_stdout(io.StringIO()):
            if fn_name == "fast3":
                out = performBeliefPropagationFast(H, syndromes[i], prior, verbose=False,
                                                   maxIter=max_iter)
            elif fn_name == "loop3":
                out = performBeliefPropagation(H, syndromes[i], prior, verbose=False,
                                               maxIter=max_iter)
            elif fn_name == "fast4":
                out = rework.performBeliefPropagationFast(H, syndromes[i], prior, maxIter=max_iter)
            elif fn_name == "minsum":
                out = rework.performMinSum_Symmetric(H, syndromes[i], prior, maxIter=max_iter, **kw)
            elif fn_name == "sym":
                out = rework.performBeliefPropagation_Symmetric(H, syndromes[i], prior,
                                                                maxIter=max_iter, **kw)
            else:
                raise KeyError(fn_name)
        hard[i], conv[i], llr[i] = out[0], out[1], out[2]
        if len(out) == 4:
            iters[i] = out[3]
    return hard, conv, iters, llr


def main():
    code_files = {"steane": "steane", "72": "[[72, 12, 6]]", "90": "[[90, 8, 10]]",
                  "108": "[[108, 8, 10]]", "144": "[[144, 12, 12]]", "288": "[[288, 12, 18]]"}
    for tag, fname in code_files.items():
        H = np.load(os.path.join(REF, "codes", f"{fname}.npz"))["Hx"]
        m, n = H.shape
        rng = np.random.default_rng(SEED)
        arrays, manifest = {}, []

        def add(fn, syndromes, prior, max_iter, errors=None, note="", **kw):
            if fn == "batch":
                with contextlib.redirect_stdout(io.StringIO()):
                    hard, conv, llr = performBeliefPropagationBatch(H, syndromes, prior,
                                                                    maxIter=max_iter)
                iters = np.full(len(conv), -1, np.int32)
            else:
                hard, conv, iters, llr = run_single(fn, H, syndromes, prior, max_iter, **kw)
            k = f"case{len(manifest):02d}"
            arrays[f"{k}/syndromes"] = syndromes.astype(np.uint8)
            arrays[f"{k}/prior"] = np.asarray(prior, np.float64)
            arrays[f"{k}/hard"] = hard.astype(np.uint8)
            arrays[f"{k}/converged"] = conv.astype(np.uint8)
            arrays[f"{k}/iters"] = iters.astype(np.int32)
            arrays[f"{k}/llr"] = llr
            if errors is not None:
                arrays[f"{k}/errors"] = errors.astype(np.uint8)
            manifest.append(dict(key=k, fn=fn, max_iter=max_iter, note=note, kw=kw,
                                 n_converged=int(conv.sum()), B=int(len(conv))))
            print(f"  {tag} {k} {fn:6s} maxIter={max_iter:3d} B={len(conv):4d} "
                  f"converged={int(conv.sum()):4d} {note} {kw}")

        small = tag in ("90", "108")
        nb = 16 if small else 32
        ps = (0.05,) if small else (0.01, 0.05, 0.10)
        if tag == "steane":
            # config 1 of BASELINE.json: Steane, p = 0.1 (main.py:17), 1k trials, 20 iterations
            e, s = generate_errors_and_syndromes_batch(H, 0.1, 1000, rng)
            add("fast4", s, prior_of(0.1, n), 20, errors=e, note="config1 p=0.1")
            add("loop3", s[:64], prior_of(0.1, n), 20, errors=e[:64], note="config1 p=0.1")
            # main.py known answer (errors on qubits 0 and 1, default maxIter=50)
            e = np.zeros((1, n), np.int64); e[0, 0] = e[0, 1] = 1
            add("loop3", (e @ H.T) % 2, prior_of(0.1, n), 50, errors=e, note="main.py")
        for p in ps:
            e, s = generate_errors_and_syndromes_batch(H, p, nb, rng)
            pr = prior_of(p, n)
            for mi in (20, 50):
                add("fast4", s, pr, mi, errors=e, note=f"p={p}")
            add("fast3", s[:8], pr, 50, errors=e[:8], note=f"p={p}")
            add("loop3", s[:4], pr, 50, errors=e[:4], note=f"p={p}")
            add("batch", s[:8], pr, 50, errors=e[:8], note=f"p={p}")
            # config-3 parameterisation (rework/Alvarado.py:153-155) with a fixed alpha
            add("minsum", s, pr, 50, errors=e, note=f"p={p}", alpha=0.8, damping=0.7, clip_llr=25.0)
            add("minsum", s[:8], pr, 50, errors=e[:8], note=f"p={p} defaults")
            add("sym", s, pr, 50, errors=e, note=f"p={p} defaults")
            add("sym", s[:8], pr, 30, errors=e[:8], note=f"p={p}", alpha=0.9, damping=0.7,
                clip_llr=25.0)
        # all-zero syndrome
        add("fast4", np.zeros((1, m), np.int8), prior_of(0.05, n), 50, note="zero syndrome")
        # single-qubit errors on every qubit (72 only) or the first 16
        nq = n if tag in ("72", "steane") else 16
        e = np.eye(n, dtype=np.int64)[:nq]
        add("fast4", (e @ H.T) % 2, prior_of(0.01, n), 50, errors=e, note="single-qubit errors")
        # deliberately non-converging: Bernoulli(0.2) errors decoded with p = 0.01 priors
        e, s = generate_errors_and_syndromes_batch(H, 0.2, 12, rng)
        add("fast4", s, prior_of(0.01, n), 50, errors=e, note="non-converging p_err=0.2")
        add("minsum", s, prior_of(0.01, n), 50, errors=e, note="non-converging p_err=0.2",
            alpha=0.8, damping=0.7, clip_llr=25.0)
        add("sym", s, prior_of(0.01, n), 50, errors=e, note="non-converging p_err=0.2")
        # non-uniform priors (studies/studyComplete.py:88-89 passes per-variable priors)
        pv = rng.uniform(0.005, 0.15, n)
        e = (rng.random((nb, n)) < pv).astype(np.int8)
        add("fast4", (e @ H.T) % 2, np.log((1 - pv) / pv), 50, errors=e, note="non-uniform priors")
        # maxIter = 1
        e, s = generate_errors_and_syndromes_batch(H, 0.05, 8, rng)
        add("fast4", s, prior_of(0.05, n), 1, errors=e, note="maxIter=1")

        arrays["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
        arrays["H"] = H.astype(np.uint8)
        path = os.path.join(HERE, f"bp_{tag}.npz")
        np.savez_compressed(path, **arrays)
        print(f"wrote {path}: {os.path.getsize(path)} bytes, {len(manifest)} cases")


def irregular():
    """Matrices beyond the (row weight 6, column weight 3) shape of codes/*.npz: the reference's
    space-time matrix (spaceTime.py:4-18, used by studies/studyTT.py:33-49) and a random sparse
    matrix with wide rows/columns and per-variable priors (studies/studyComplete.py:88-89)."""
    sys.path.insert(0, REF)
    from spaceTime import spaceTimeMatrix, spacetimeSyndrome
    H72 = np.load(os.path.join(REF, "codes", "[[72, 12, 6]].npz"))["Hx"]
    cases = {}
    np.random.seed(20260128)
    Hst = spaceTimeMatrix(H72, 3)                      # (108, 324) float 0/1
    cases["st72"] = (Hst, [spacetimeSyndrome(H72, 0.02, 3)[1] for _ in range(12)]
                     + [spacetimeSyndrome(H72, 0.06, 3)[1] for _ in range(8)], None)
    rng = np.random.default_rng(SEED)
    m, n = 60, 120
    Hr = np.zeros((m, n), np.int64)
    for c in range(m):
        Hr[c, rng.choice(n, rng.integers(2, 13), replace=False)] = 1
    Hr[:, 5] = 0                                       # an isolated variable
    Hr[7, :] = 0                                       # an empty check
    pv = rng.uniform(0.01, 0.2, n)
    e = (rng.random((24, n)) < pv * 0.5).astype(np.int64)
    cases["rand"] = (Hr, list((e @ Hr.T) % 2), np.log((1 - pv) / pv))
    for tag, (H, syns, prior) in cases.items():
        m, n = H.shape
        syns = np.array(syns).astype(np.int64)
        arrays, manifest = {}, []
        pr = prior if prior is not None else prior_of(0.03, n)
        for fn, mi, kw in (("fast4", 50, {}), ("fast4", 5, {}),
                           ("minsum", 50, dict(alpha=0.8, damping=0.7, clip_llr=25.0)),
                           ("sym", 50, {})):
            hard, conv, iters, llr = run_single(fn, H, syns, pr, mi, **kw)
            k = f"case{len(manifest):02d}"
            arrays[f"{k}/syndromes"] = syns.astype(np.uint8)
            arrays[f"{k}/prior"] = np.asarray(pr, np.float64)
            arrays[f"{k}/hard"] = hard.astype(np.uint8)
            arrays[f"{k}/converged"] = conv.astype(np.uint8)
            arrays[f"{k}/iters"] = iters.astype(np.int32)
            arrays[f"{k}/llr"] = llr
            manifest.append(dict(key=k, fn=fn, max_iter=mi, note=tag, kw=kw,
                                 n_converged=int(conv.sum()), B=int(len(conv))))
            print(f"  {tag} {k} {fn:6s} maxIter={mi:3d} B={len(conv):3d} converged={int(conv.sum()):3d} "
                  f"row wt<= {int(H.sum(1).max())} col wt<= {int(H.sum(0).max())}")
        arrays["manifest"] = np.frombuffer(json.dumps(manifest).encode(), dtype=np.uint8)
        arrays["H"] = (np.asarray(H) != 0).astype(np.uint8)
        path = os.path.join(HERE, f"bp_{tag}.npz")
        np.savez_compressed(path, **arrays)
        print(f"wrote {path}: {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    if "--irregular-only" not in sys.argv:
        main()
    irregular()
