"""Mirror of ``rework/decoding.py`` (the 4-tuple decoders that also return the iteration).

* performBeliefPropagationFast          rework/decoding.py:77-129
* performMinSum_Symmetric               rework/decoding.py:5-75
* performBeliefPropagation_Symmetric    rework/decoding.py:131-191
* performOSD_enhanced                   rework/decoding.py:193 (see qldpc_amd/osd.py)

``alpha_estimation=True`` (the message dump behind rework/Alvarado.py:10-66) returns, like the
reference, ``(0, 0, R, 0)`` with R the dense (m, n) matrix of check->variable messages
(min-sum: ``R_new / alpha`` after the first check update, :58-59; damped sum-product: ``R`` at
iteration 10, :168-169); ``qldpc_amd.alvarado.estimate_alpha_from_code`` is the batched form.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .bp import _check_iter, _prior, _syndromes, decode_one, decoder_for, dense_colsum_flags
from .osd import performOSD_enhanced  # noqa: F401  (rework/decoding.py:193; rework/main.py:6 imports it)


def _dense_messages(H, syndrome, initialBelief, variant, alpha, damping, clip_llr, iteration):
    dec = decoder_for(H)
    syn = _syndromes(syndrome, dec.m, batch=False).astype(np.uint8)
    msgs = dec.check_messages(syn[None, :], _prior(initialBelief, dec.n), variant, alpha, damping,
                              clip_llr, iteration, flags=dense_colsum_flags(H, damped=True))[0]
    R = np.zeros((dec.m, dec.n))
    rows = np.repeat(np.arange(dec.m), np.diff(dec.row_ptr))
    R[rows, dec.col_idx] = msgs
    return R


def performBeliefPropagationFast(H, syndrome, initialBelief, maxIter=50):
    hard, conv, llr, it = decode_one(H, syndrome, initialBelief, maxIter, _lib.SUM_PRODUCT,
                                     flags=dense_colsum_flags(H))
    return hard, conv, llr, it


def performMinSum_Symmetric(H, syndrome, initialBelief, maxIter=50, alpha=1.0, damping=1.0,
                            clip_llr=20.0, alpha_estimation=False):
    if alpha_estimation:                                      # rework/decoding.py:58-59
        _check_iter(maxIter)
        return 0, 0, _dense_messages(H, syndrome, initialBelief, _lib.MIN_SUM, alpha, damping,
                                     clip_llr, 0), 0
    hard, conv, llr, it = decode_one(H, syndrome, initialBelief, maxIter, _lib.MIN_SUM, alpha,
                                     damping, clip_llr, flags=dense_colsum_flags(H, damped=True))
    return hard, conv, llr, it


def performBeliefPropagation_Symmetric(H, syndrome, initialBelief, maxIter=50, alpha=1.0,
                                       damping=0.8, clip_llr=20.0, alpha_estimation=False):
    if alpha_estimation:                                      # rework/decoding.py:168-169
        if _check_iter(maxIter) <= 10:
            raise NotImplementedError("alpha_estimation=True needs maxIter > 10 (the reference only "
                                      "returns messages at iteration 10)")
        return 0, 0, _dense_messages(H, syndrome, initialBelief, _lib.DAMPED_SP, alpha, damping,
                                     clip_llr, 10), 0
    hard, conv, llr, it = decode_one(H, syndrome, initialBelief, maxIter, _lib.DAMPED_SP, alpha,
                                     damping, clip_llr, flags=dense_colsum_flags(H, damped=True))
    return hard, conv, llr, it
