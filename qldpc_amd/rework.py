"""Mirror of ``rework/decoding.py`` (the 4-tuple decoders that also return the iteration).

* performBeliefPropagationFast          rework/decoding.py:77-129
* performMinSum_Symmetric               rework/decoding.py:5-75
* performBeliefPropagation_Symmetric    rework/decoding.py:131-191

``alpha_estimation=True`` (the message-dump mode behind rework/Alvarado.py:10-66) is not part
of the accelerated path yet and raises ``NotImplementedError``.
"""
from __future__ import annotations

from . import _lib
from .bp import decode_one


def performBeliefPropagationFast(H, syndrome, initialBelief, maxIter=50):
    hard, conv, llr, it = decode_one(H, syndrome, initialBelief, maxIter, _lib.SUM_PRODUCT)
    return hard, conv, llr, it


def performMinSum_Symmetric(H, syndrome, initialBelief, maxIter=50, alpha=1.0, damping=1.0,
                            clip_llr=20.0, alpha_estimation=False):
    if alpha_estimation:
        raise NotImplementedError("alpha_estimation message dump is not accelerated (SURVEY 8(f) #4)")
    hard, conv, llr, it = decode_one(H, syndrome, initialBelief, maxIter, _lib.MIN_SUM, alpha,
                                     damping, clip_llr)
    return hard, conv, llr, it


def performBeliefPropagation_Symmetric(H, syndrome, initialBelief, maxIter=50, alpha=1.0,
                                       damping=0.8, clip_llr=20.0, alpha_estimation=False):
    if alpha_estimation:
        raise NotImplementedError("alpha_estimation message dump is not accelerated (SURVEY 8(f) #4)")
    hard, conv, llr, it = decode_one(H, syndrome, initialBelief, maxIter, _lib.DAMPED_SP, alpha,
                                     damping, clip_llr)
    return hard, conv, llr, it
