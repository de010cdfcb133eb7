#!/usr/bin/env python3
"""Golden vectors for the alpha_estimation=True message dumps and the alpha fit, produced by
RUNNING THE REAL REFERENCE decoders (build container only):

    MPLBACKEND=Agg python tests/golden/make_golden_alpha.py

  rework/decoding.py:5    performMinSum_Symmetric(..., alpha_estimation=True)  -> R_new/alpha, iteration 0
  rework/decoding.py:131  performBeliefPropagation_Symmetric(..., alpha_estimation=True) -> R, iteration 10
The alpha value follows the procedure of rework/Alvarado.py:10-66 (that file cannot be imported:
its module body runs the whole experiment), driven by the reference's decoder.
"""
import importlib.util
import os

import numpy as np
from scipy.optimize import curve_fit

REF = os.environ.get("QLDPC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_rework_decoding", os.path.join(REF, "rework", "decoding.py"))
rework = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rework)


def main():
    H = np.load(os.path.join(REF, "codes", "[[72, 12, 6]].npz"))["Hx"]
    m, n = H.shape
    rows, cols = np.nonzero(H)
    rng = np.random.default_rng(20260128)
    p = 0.05
    prior = np.array([np.log((1 - p) / p)] * n)
    e = (rng.random((8, n)) < p).astype(int)
    syn = (e @ H.T) % 2
    pv = np.log((1 - 0.05) / 0.05) * rng.uniform(0.5, 1.5, n)      # non-uniform priors
    out = {"H": H.astype(np.uint8), "syndromes": syn.astype(np.uint8), "prior": prior, "prior_nu": pv}
    ms, ms_nu, sp = [], [], []
    for s in syn:
        ms.append(rework.performMinSum_Symmetric(H, s, prior, maxIter=1, alpha=1.0, damping=1.0,
                                                 clip_llr=np.inf, alpha_estimation=True)[2][rows, cols])
        ms_nu.append(rework.performMinSum_Symmetric(H, s, pv, maxIter=50, alpha=0.8, damping=0.7,
                                                    clip_llr=25.0, alpha_estimation=True)[2][rows, cols])
        sp.append(rework.performBeliefPropagation_Symmetric(H, s, pv, maxIter=50, alpha=0.9, damping=0.8,
                                                            clip_llr=20.0, alpha_estimation=True)[2][rows, cols])
    out["minsum_R"] = np.array(ms); out["minsum_nu_R"] = np.array(ms_nu); out["sym_R"] = np.array(sp)
    # alpha fit, rework/Alvarado.py:10-66 with the reference decoder; non-uniform channel so that the
    # message histogram is not two spikes: error_rate 0.08, 400 trials, seed 123
    np.random.seed(123)
    trials, error_rate, bins = 400, 0.08, 50
    true_0, true_1 = [], []
    for _ in range(trials):
        beliefs = [np.log((1 - error_rate) / error_rate)] * n
        error = (np.random.random(n) < error_rate).astype(int)
        syndrome = (error @ H.T) % 2
        _, _, R, _ = rework.performMinSum_Symmetric(H, syndrome, beliefs, maxIter=1, alpha=1.0, damping=1.0,
                                                     clip_llr=np.inf, alpha_estimation=True)
        vm = R[rows, cols]
        bits = error[cols]
        true_0.extend(vm[bits == 0]); true_1.extend(vm[bits == 1])
    true_0, true_1 = np.array(true_0), np.array(true_1)
    rng_h = (min(true_0.min(), true_1.min()), max(true_0.max(), true_1.max()))
    h0, edges = np.histogram(true_0, bins=bins, range=rng_h, density=True)
    h1, _ = np.histogram(true_1, bins=bins, range=rng_h, density=True)
    centers = (edges[:-1] + edges[1:]) / 2
    valid = (h0 > 0) & (h1 > 0)
    popt, _ = curve_fit(lambda x, a: a * x, centers[valid], np.log(h0[valid] / h1[valid]))
    out["alpha_fit"] = np.array([popt[0]])
    out["alpha_fit_args"] = np.array([trials, error_rate, bins, 123])
    print("alpha", popt[0], "valid bins", int(valid.sum()))
    path = os.path.join(HERE, "alpha_est.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
