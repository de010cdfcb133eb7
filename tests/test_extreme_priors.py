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
