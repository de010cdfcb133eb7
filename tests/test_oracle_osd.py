"""CPU: the oracle's OSD-0 (oracle/bp_oracle.c:oracle_osd0) against the reference's performOSD
outputs (tests/golden/osd.npz, made by make_golden_osd.py)."""
import os

import numpy as np
import pytest

from oracle import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "osd.npz")
TAGS = ("steane", "72", "144", "288")


def load_osd(tag):
    d = np.load(GOLD)
    return {k: d[f"{tag}/{k}"] for k in ("H", "syndromes", "llr", "hard", "solution", "kind")}


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_osd0_matches_reference(tag):
    c = load_osd(tag)
    H = c["H"].astype(np.int64)
    for s, l, h, want in zip(c["syndromes"], c["llr"], c["hard"], c["solution"]):
        got = oracle.osd0(H, s, l, h)
        assert np.array_equal((got.astype(np.int64) @ H.T) % 2, s)
        # Distinct reliabilities: bit-exact by construction.  Tied |llr| (numpy's unstable
        # argsort decides the reference's order): equal in every committed vector as well.
        assert np.array_equal(got, want)


@pytest.mark.parametrize("tag", ("72", "144", "288"))
def test_oracle_osd0_on_syndromes_outside_the_column_space(tag):
    """tests/golden/osd_inconsistent.npz: the reference's performOSD on random syndromes (none of them in the
    column space).  Its output then depends on the row swaps of gf2_elimination (OSD.py:56-59), which the oracle
    follows."""
    d = np.load(os.path.join(os.path.dirname(GOLD), "osd_inconsistent.npz"))
    H = d[f"{tag}/H"].astype(np.int64)
    assert not d[f"{tag}/reproduces_syndrome"].any()
    for s, l, h, want in zip(d[f"{tag}/syndromes"], d[f"{tag}/llr"], d[f"{tag}/hard"], d[f"{tag}/solution"]):
        assert np.array_equal(oracle.osd0(H, s, l, h), want)
