"""``estimate_alpha_from_code`` of rework/Alvarado.py:10-66 -- the min-sum normalisation alpha
fitted to the distribution of first-iteration check->variable messages -- with the heavy part on
the MI355X.

What the reference does, and where each step runs here:

1. ``trials`` random errors and their syndromes (:17-21).  Host, drawn from ``np.random.random``
   exactly as the reference draws them, so ``np.random.seed`` reproduces its estimate.
2. One min-sum check update per trial with ``alpha_estimation=True`` (:27-29), messages split by the
   true value of the addressed bit (:31-36), one 50-bin histogram per class over the common range
   (:41-47).  Device: ``qbp_message_histograms`` -- the ``trials * E`` messages are produced, ranged
   and binned in HBM / LDS and only ``2 * bins`` counters come back.
3. Densities, log-ratio on the bins both classes populate, and the slope of the line through the
   origin that fits it best (:46-62).  Host, below: for the one-parameter model ``f = alpha * x``
   the least-squares solution is ``sum(x f) / sum(x x)``; the reference reaches the same number
   through ``scipy.optimize.curve_fit``.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .bp import decoder_for


def slope_through_origin(x, f):
    """argmin_a sum (f - a x)^2."""
    x = np.asarray(x, np.float64)
    f = np.asarray(f, np.float64)
    xx = float(np.dot(x, x))
    if x.size == 0 or xx == 0.0:
        # (the reference's curve_fit raises on an empty / degenerate data set; a silent 0 / 0 = NaN here
        # would be handed on as the min-sum normalisation)
        raise ValueError("alpha fit: no bin is populated by both message classes (nothing to fit)")
    return float(np.dot(x, f) / xx)


def alpha_from_histograms(edges, count0, count1):
    """alpha from the two raw histograms: density = count / (total * bin width) (np.histogram's
    density=True), log ratio where both are positive, slope through the origin at the bin centres."""
    widths = np.diff(edges)
    d0 = count0 / widths / count0.sum()
    d1 = count1 / widths / count1.sum()
    both = (d0 > 0) & (d1 > 0)
    centres = (edges[:-1] + edges[1:]) / 2
    return slope_through_origin(centres[both], np.log(d0[both] / d1[both]))


def estimate_alpha_from_code(code, trials=5000, error_rate=0.05, maxIter=50, bins=50):
    code = np.asarray(code)
    n = code.shape[1]
    dec = decoder_for(code)
    prior = np.full(n, np.log((1 - error_rate) / error_rate))
    errors = np.empty((trials, n), np.uint8)
    for t in range(trials):                       # one call per trial: the reference's draw order
        errors[t] = np.random.random(n) < error_rate
    syndromes = (errors.astype(np.int64) @ code.T % 2).astype(np.uint8)
    edges, count0, count1 = dec.message_histograms(syndromes, errors, prior, _lib.MIN_SUM, alpha=1.0,
                                                   damping=1.0, clip_llr=np.inf, iteration=0, bins=bins)
    if count0.sum() == 0 or count1.sum() == 0:
        raise ValueError("one of the two message classes is empty (the reference fails there too)")
    alpha = alpha_from_histograms(edges, count0, count1)
    print(f"Estimated alpha for error rate {error_rate}: {alpha}")     # :64
    return alpha
