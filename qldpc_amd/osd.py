"""Mirror of ``decoding/OSD.py``: ``performOSD(H, syndrome, llr, hard)`` (OSD-0), on the GPU.

Same positional signature and return type (int64 vector, ``(hard + e_correction) % 2``,
OSD.py:26-28).  ``decoding/OSD_enhanced.py`` with ``order=0`` computes the same thing; higher
orders are not accelerated.  Equal ``|llr|`` values are ordered by column index (the reference's
``np.argsort`` leaves that order to the numpy build).
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
