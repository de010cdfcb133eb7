"""Mirror of ``decoding/OSD.py``: ``performOSD(H, syndrome, llr, hard)`` (OSD-0), on the GPU.

Same positional signature and return type (int64 vector, ``(hard + e_correction) % 2``,
OSD.py:26-28).  Equal ``|llr|`` values are ordered by column index (the reference's ``np.argsort``
leaves that order to the numpy build).

``performOSD_enhanced`` (decoding/OSD_enhanced.py:5 = rework/decoding.py:193) returns its OSD-0
solution whenever that solution reproduces the syndrome (OSD_enhanced.py "if np.all(osd0_syndrome
== syndrome): return osd0_solution", before any higher-order search) -- which is the case for every
syndrome in the column space of H, i.e. every syndrome that comes from an error.  The mirror below
therefore is OSD-0 on the GPU for every ``order``; only an INCONSISTENT syndrome with ``order > 0``
would reach the reference's combinatorial search, which is not implemented here.
"""
from __future__ import annotations

import numpy as np

from .bp import decoder_for


def performOSD(H, syndrome, llr, hard):
    dec = decoder_for(H)
    syn = (np.asarray(syndrome).astype(np.int64) % 2).astype(np.uint8)
    hd = (np.asarray(hard).astype(np.int64) % 2).astype(np.uint8)
    l = np.asarray(llr, dtype=np.float64)
    if syn.shape != (dec.m,) or hd.shape != (dec.n,) or l.shape != (dec.n,):
        raise ValueError(f"expected syndrome ({dec.m},), llr ({dec.n},), hard ({dec.n},)")
    return dec.osd0(syn[None, :], l[None, :], hd[None, :])[0].astype(np.int64)


def performOSD_batch(H, syndromes, llrs, hards):
    """Batch form (no reference counterpart): uint8[B, n] solutions."""
    return decoder_for(H).osd0(syndromes, llrs, hards)


def performOSD_enhanced(H, syndrome, llr, hard, order=0, max_combinations=None):
    sol = performOSD(H, syndrome, llr, hard)
    if order == 0:
        return sol
    from scipy.sparse import issparse
    Hd = H.toarray() if issparse(H) else np.asarray(H)
    if np.array_equal((sol @ (Hd != 0).astype(np.int64).T) % 2, np.asarray(syndrome).astype(np.int64) % 2):
        return sol                                   # the reference returns here as well
    raise NotImplementedError("performOSD_enhanced(order > 0) on a syndrome outside the column space "
                              "of H: the reference's combinatorial search is not implemented")
