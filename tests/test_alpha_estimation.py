"""alpha_estimation=True message dumps (rework/decoding.py:58-59, :168-169) and the alpha fit of
rework/Alvarado.py:10-66: oracle on CPU, device + Python mirrors on GPU, against vectors produced
by the reference's decoders (tests/golden/alpha_est.npz)."""
import os

import numpy as np
import pytest

from oracle import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "alpha_est.npz")


def gold():
    return np.load(GOLD)


def test_oracle_message_dumps_match_reference():
    d = gold()
    H = d["H"].astype(np.int64)
    a = oracle.check_messages(H, d["syndromes"], d["prior"], 2, 1.0, 1.0, np.inf, 0)
    assert np.array_equal(a, d["minsum_R"])                  # no transcendental: bit-exact
    b = oracle.check_messages(H, d["syndromes"], d["prior_nu"], 2, 0.8, 0.7, 25.0, 0)
    assert np.array_equal(b, d["minsum_nu_R"])
    # (the golden's H is the Fortran-ordered Hx of the code file: column sums of the ten iterations before
    # the dump in the reference's order, oracle.colsum_flags)
    c = oracle.check_messages(H, d["syndromes"], d["prior_nu"], 1, 0.9, 0.8, 20.0, 10,
                              flags=oracle.colsum_flags("sym", H))
    assert np.array_equal(c, d["sym_R"])                     # numpy's tanh / arctanh kernels: bit-exact too


@pytest.mark.gpu
def test_device_message_dumps_and_alpha_fit(capsys):
    from qldpc_amd import _lib, alvarado, bp, rework
    d = gold()
    H = d["H"].astype(np.int64)
    dec = bp.decoder_for(H)
    a = dec.check_messages(d["syndromes"], d["prior"], _lib.MIN_SUM, 1.0, 1.0, np.inf, 0)
    assert np.array_equal(a, d["minsum_R"])
    b = dec.check_messages(d["syndromes"], d["prior_nu"], _lib.MIN_SUM, 0.8, 0.7, 25.0, 0)
    assert np.array_equal(b, d["minsum_nu_R"])
    c = dec.check_messages(d["syndromes"], d["prior_nu"], _lib.DAMPED_SP, 0.9, 0.8, 20.0, 10,
                           flags=bp.dense_colsum_flags(H, damped=True))
    assert np.array_equal(c, d["sym_R"])
    # reference-shaped return value: (0, 0, dense R, 0)
    out = rework.performMinSum_Symmetric(H, d["syndromes"][0], list(d["prior"]), maxIter=1, alpha=1.0,
                                         damping=1.0, clip_llr=np.inf, alpha_estimation=True)
    rows, cols = np.nonzero(H)
    assert out[0] == 0 and out[1] == 0 and out[3] == 0 and out[2].shape == H.shape
    assert np.array_equal(out[2][rows, cols], d["minsum_R"][0]) and out[2][H == 0].sum() == 0
    # the alpha fit with the reference's RNG stream
    trials, error_rate, bins, seed = d["alpha_fit_args"]
    np.random.seed(int(seed))
    alpha = alvarado.estimate_alpha_from_code(H, trials=int(trials), error_rate=float(error_rate),
                                              maxIter=1, bins=int(bins))
    assert "Estimated alpha for error rate" in capsys.readouterr().out
    assert alpha == pytest.approx(float(d["alpha_fit"][0]), rel=1e-12)


def test_closed_form_fit_is_the_references_fit():
    """qldpc_amd.alvarado.alpha_from_histograms (density, log ratio, slope through the origin) against
    the reference's own sequence of numpy / scipy calls (rework/Alvarado.py:41-62) on synthetic
    messages: np.histogram(..., density=True) on a common range, curve_fit of alpha * x."""
    from scipy.optimize import curve_fit

    from qldpc_amd import alvarado
    rng = np.random.default_rng(11)
    for _ in range(5):
        m0 = rng.normal(2.0, 1.5, 40000)          # messages about bits that are 0 / 1
        m1 = rng.normal(-2.0, 1.5, 3000)
        rng_ = (min(m0.min(), m1.min()), max(m0.max(), m1.max()))
        h0, edges = np.histogram(m0, bins=50, range=rng_, density=True)
        h1, _ = np.histogram(m1, bins=50, range=rng_, density=True)
        centres = (edges[:-1] + edges[1:]) / 2
        ok = (h0 > 0) & (h1 > 0)
        want = curve_fit(lambda x, a: a * x, centres[ok], np.log(h0[ok] / h1[ok]))[0][0]
        c0, _ = np.histogram(m0, bins=50, range=rng_)
        c1, _ = np.histogram(m1, bins=50, range=rng_)
        got = alvarado.alpha_from_histograms(edges, c0, c1)
        assert got == pytest.approx(want, rel=1e-9)


def test_fit_without_common_bins_raises_like_the_reference():
    """when no histogram bin holds messages of both classes the reference's curve_fit raises
    (rework/Alvarado.py:55-62); the closed form must not return NaN instead."""
    from qldpc_amd import alvarado
    edges = np.linspace(-1.0, 1.0, 5)
    with pytest.raises(ValueError, match="both message classes"):
        alvarado.alpha_from_histograms(edges, np.array([5, 3, 0, 0]), np.array([0, 0, 2, 7]))
    with pytest.raises(ValueError):
        alvarado.slope_through_origin([0.0, 0.0], [1.0, 2.0])
