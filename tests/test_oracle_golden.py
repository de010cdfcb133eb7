"""CPU: the oracle (oracle/bp_oracle.c) against vectors produced by the real reference."""
import numpy as np
import pytest

import golden_util
from oracle import oracle


@pytest.mark.parametrize("tag", golden_util.TAGS + golden_util.IRREGULAR_TAGS)
def test_oracle_matches_reference_goldens(tag):
    n_cases = 0
    worst = (0.0, 0.0)
    for case in golden_util.load(tag):
        hard, conv, iters, llr = oracle.decode_batch(
            case["H"], case["syndromes"], case["prior"], case["max_iter"], case["variant"],
            case["alpha"], case["damping"], case["clip_llr"])
        w = golden_util.compare(case, hard, conv, iters, llr, "oracle")
        worst = (max(worst[0], w[0]), max(worst[1], w[1]))
        n_cases += 1
    assert n_cases >= (4 if tag in golden_util.IRREGULAR_TAGS else 10)
    print(f"{tag}: {n_cases} cases, worst LLR rel err converged {worst[0]:.2e}, "
          f"non-converged {worst[1]:.2e}")


def test_oracle_force_full_freezes_outputs():
    case = next(c for c in golden_util.load("72") if c["fn"] == "fast4" and c["max_iter"] == 50)
    a = oracle.decode_batch(case["H"], case["syndromes"], case["prior"], 50)
    b = oracle.decode_batch(case["H"], case["syndromes"], case["prior"], 50,
                            flags=oracle.FLAG_FORCE_FULL)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_main_py_known_answer():
    # main.py on Steane, errors on qubits 0 and 1, p = 0.1
    case = next(c for c in golden_util.load("steane") if c["note"] == "main.py")
    hard, conv, iters, llr = oracle.decode_batch(case["H"], case["syndromes"], case["prior"], 50)
    assert hard[0].tolist() == [0, 0, 1, 0, 0, 0, 0] and conv[0] and iters[0] == 0
    np.testing.assert_allclose(
        llr[0], [1.06635143, 1.06635143, -0.06452172, 3.32809773, 2.19722458, 2.19722458,
                 1.06635143], rtol=1e-7)
