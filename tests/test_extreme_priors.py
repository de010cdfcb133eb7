"""Extreme priors (0, negative, +-inf, saturating, |prior| < 2e-15) against vectors produced by
the real reference (tests/golden/make_golden_extreme.py): CPU oracle here, HIP path under -m gpu."""
import numpy as np
import pytest

import golden_util
from oracle import oracle


@pytest.mark.parametrize("tag", golden_util.EXTREME_TAGS)
def test_oracle_extreme_priors(tag):
    n_cases, worst = 0, (0.0, 0)
    for case in golden_util.load(tag):
        with np.errstate(all="ignore"):
            out = oracle.decode_batch(case["H"], case["syndromes"], case["prior"], case["max_iter"],
                                      case["variant"], case["alpha"], case["damping"], case["clip_llr"])
            w = golden_util.compare_extreme(case, *out, "oracle")
        worst = (max(worst[0], w[0]), worst[1] + w[1])
        n_cases += 1
    assert n_cases == 54
    print(f"{tag}: worst LLR error on compared syndromes {worst[0]:.2e}; "
          f"hard-decision flips on chaotic non-converged syndromes: {worst[1]}")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", golden_util.EXTREME_TAGS)
@pytest.mark.parametrize("kernel", [0, 2, 3])
def test_hip_extreme_priors(tag, kernel):
    from qldpc_amd import _lib, bp
    worst = (0.0, 0)
    for case in golden_util.load(tag):
        dec = bp.decoder_for(case["H"])
        dec.set_option(_lib.OPT_KERNEL, kernel)     # 0: on-chip for x72, general-H for xrand
        try:
            out = dec.decode(case["syndromes"], case["prior"], case["max_iter"], case["variant"],
                             case["alpha"], case["damping"], case["clip_llr"])
        finally:
            dec.set_option(_lib.OPT_KERNEL, 0)
        with np.errstate(all="ignore"):
            w = golden_util.compare_extreme(case, *out, f"hip kernel {kernel}")
        worst = (max(worst[0], w[0]), worst[1] + w[1])
    print(f"{tag} kernel {kernel}: worst LLR error {worst[0]:.2e}; flips on chaotic syndromes {worst[1]}")


# damping = 1.0 with infinite priors: 1 * inf + 0 * inf = NaN messages in the damped variant
# (tests/golden/make_golden_damping1.py; ADVICE r01: the device used to clip those NaNs away)
DAMP1_TAGS = ("xd72", "xdrand")


def _compare_nan_case(case, hard, conv, iters, llr, who):
    name = f"{who} {case['tag']}/{case['key']} {case['note']} {case['max_iter']}"
    assert np.array_equal(conv, case["converged"]), name
    assert np.array_equal(iters, case["iters"]), name
    ref = case["llr"]
    assert np.array_equal(np.isnan(llr), np.isnan(ref)), f"NaN pattern differs: {name}"
    inf = np.isinf(ref)
    assert np.array_equal(np.isinf(llr), inf) and np.array_equal(llr[inf], ref[inf]), name
    # values: on converged syndromes (a syndrome still iterating may sit on an exact cancellation that
    # the reference scales by 1e15, see make_golden_extreme.py; NaN / inf patterns above: on all)
    strict = case["converged"] & ~case["noisy"]
    fin = np.isfinite(ref) & strict[:, None]
    if fin.any():
        d = np.abs(llr[fin] - ref[fin])
        assert (d <= np.maximum(1e-5 * np.abs(ref[fin]), 1e-7)).all(), f"{d.max():.2e}: {name}"
    assert np.array_equal(hard[strict], case["hard"][strict]), f"hard differs: {name}"


@pytest.mark.parametrize("tag", DAMP1_TAGS)
def test_oracle_damping_one_nan_path(tag):
    n = 0
    for case in golden_util.load(tag):
        with np.errstate(all="ignore"):
            out = oracle.decode_batch(case["H"], case["syndromes"], case["prior"], case["max_iter"],
                                      case["variant"], case["alpha"], case["damping"], case["clip_llr"])
        _compare_nan_case(case, *out, "oracle")
        n += 1
    assert n == 9


@pytest.mark.gpu
@pytest.mark.parametrize("tag", DAMP1_TAGS)
@pytest.mark.parametrize("kernel", [0, 2, 3])
def test_hip_damping_one_nan_path(tag, kernel):
    from qldpc_amd import _lib, bp
    for case in golden_util.load(tag):
        dec = bp.decoder_for(case["H"])
        dec.set_option(_lib.OPT_KERNEL, kernel)
        try:
            out = dec.decode(case["syndromes"], case["prior"], case["max_iter"], case["variant"],
                             case["alpha"], case["damping"], case["clip_llr"])
        finally:
            dec.set_option(_lib.OPT_KERNEL, 0)
        _compare_nan_case(case, *out, f"hip kernel {kernel}")
