"""Mirror of ``estimate_alpha_from_code`` (rework/Alvarado.py:10-66): fit the min-sum
normalisation alpha from the distribution of first-iteration check->variable messages.

The reference runs ``trials`` one-iteration min-sum decodes in a Python loop and collects the
dense message matrices; here all trials go through ONE device call (``qbp_check_messages``), the
histogram and the one-parameter least-squares fit are the reference's, on the host.  Errors are
drawn from ``np.random.random`` exactly as the reference does (:19), so with the same
``np.random.seed`` the estimate is the reference's.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .bp import decoder_for


def estimate_alpha_from_code(code, trials=5000, error_rate=0.05, maxIter=50, bins=50):
    code = np.asarray(code)
    n = len(code[0])
    dec = decoder_for(code)
    edge_cols = dec.col_idx                                   # np.nonzero(code): CSR edge order
    initialBeliefs = np.array([np.log((1 - error_rate) / error_rate)] * n)
    errors = np.empty((trials, n), np.int64)
    for t in range(trials):                                   # same draw order as :19
        errors[t] = (np.random.random(n) < error_rate).astype(int)
    syndromes = (errors @ code.T) % 2                         # :21
    # performMinSum_Symmetric(..., alpha=1.0, damping=1.0, clip_llr=inf, alpha_estimation=True) :27-29
    R = dec.check_messages(syndromes.astype(np.uint8), initialBeliefs, _lib.MIN_SUM, alpha=1.0,
                           damping=1.0, clip_llr=np.inf, iteration=0)
    bits = errors[:, edge_cols]                               # :33
    true_0 = R[bits == 0]                                     # :35-36 (trial-major, edge order)
    true_1 = R[bits == 1]
    min_val = min(true_0.min(), true_1.min())
    max_val = max(true_0.max(), true_1.max())
    hist_range = (min_val, max_val)
    hist_0, bin_edges = np.histogram(true_0, bins=bins, range=hist_range, density=True)
    hist_1, _ = np.histogram(true_1, bins=bins, range=hist_range, density=True)
    bin_centers = (bin_edges[:-1] + bin_edges[1:]) / 2
    valid = (hist_0 > 0) & (hist_1 > 0)
    lambdas = bin_centers[valid]
    f_lambdas = np.log(hist_0[valid] / hist_1[valid])
    from scipy.optimize import curve_fit

    def linear_model(x, alpha):
        return alpha * x

    popt, _ = curve_fit(linear_model, lambdas, f_lambdas)
    alpha_opt = popt[0]
    print(f"Estimated alpha for error rate {error_rate}: {alpha_opt}")     # :64
    return alpha_opt
