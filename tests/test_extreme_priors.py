"""Extreme priors (0, negative, +-inf, saturating, |prior| < 2e-15) and the damped variant's NaN path
(damping = 1 with infinite priors: 1 * inf + 0 * inf) against vectors produced by the real reference
(tests/golden/make_golden_extreme.py, make_golden_damping1.py): CPU oracle here, HIP path under -m gpu.

Round 3: everything is compared bit for bit -- NaN / inf patterns, hard decisions, posterior values, the
syndromes still iterating at the limit, and the `noisy` ones whose reference values are numpy's own
last-ulp rounding scaled up by 1e15 (an exact cancellation divided by the +1e-15 of beliefPropagation.py:122):
oracle and device evaluate numpy's own tanh / arctanh kernels, so they reproduce those too."""
import numpy as np
import pytest

import golden_util
from oracle import oracle

DAMP1_TAGS = ("xd72", "xdrand")
N_CASES = {"x72": 54, "xrand": 54, "xd72": 9, "xdrand": 9}


@pytest.mark.parametrize("tag", golden_util.EXTREME_TAGS + DAMP1_TAGS)
def test_oracle_extreme_priors(tag):
    n_cases = n_syn = 0
    for case in golden_util.load(tag):
        with np.errstate(all="ignore"):
            out = oracle.decode_batch(case["H"], case["syndromes"], case["prior"], case["max_iter"],
                                      case["variant"], case["alpha"], case["damping"], case["clip_llr"],
                                      flags=golden_util.oracle_flags(case))
            n_syn += golden_util.compare_extreme(case, *out, "oracle")
        n_cases += 1
    assert n_cases == N_CASES[tag]
    print(f"{tag}: {n_cases} cases, {n_syn} syndromes identical to the reference in every bit")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", golden_util.EXTREME_TAGS + DAMP1_TAGS)
@pytest.mark.parametrize("kernel", [0, 2, 3])
def test_hip_extreme_priors(tag, kernel):
    from qldpc_amd import _lib, bp
    n_syn = 0
    for case in golden_util.load(tag):
        dec = bp.decoder_for(case["H"])
        dec.set_option(_lib.OPT_KERNEL, kernel)     # 0: on-chip for x72, general-H for xrand
        try:
            out = dec.decode(case["syndromes"], case["prior"], case["max_iter"], case["variant"],
                             case["alpha"], case["damping"], case["clip_llr"], golden_util.device_flags(case))
        finally:
            dec.set_option(_lib.OPT_KERNEL, 0)
        with np.errstate(all="ignore"):
            n_syn += golden_util.compare_extreme(case, *out, f"hip kernel {kernel}")
    print(f"{tag} kernel {kernel}: {n_syn} syndromes identical to the reference in every bit")
