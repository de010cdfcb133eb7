"""CPU: the oracle (oracle/bp_oracle.c) against vectors produced by the real reference."""
import numpy as np
import pytest

import golden_util
from oracle import oracle


@pytest.mark.parametrize("tag", golden_util.TAGS + golden_util.IRREGULAR_TAGS)
def test_oracle_matches_reference_goldens(tag):
    n_cases = n_syn = 0
    for case in golden_util.load(tag):
        hard, conv, iters, llr = oracle.decode_batch(
            case["H"], case["syndromes"], case["prior"], case["max_iter"], case["variant"],
            case["alpha"], case["damping"], case["clip_llr"], flags=golden_util.oracle_flags(case))
        n_syn += golden_util.compare(case, hard, conv, iters, llr, "oracle")
        n_cases += 1
    assert n_cases >= (4 if tag in golden_util.IRREGULAR_TAGS else 10)
    print(f"{tag}: {n_cases} cases, {n_syn} syndromes: hard decision, converged flag, iteration and LLR bits "
          "identical to the reference")


def test_libm_math_is_the_other_implementation():
    """ORACLE_FLAG_LIBM_MATH (host libm tanh / atanh: what numpy itself computes on hosts without AVX512_SKX)
    keeps hard decisions / flags / iterations on these vectors and moves LLRs in the last digits only."""
    case = next(c for c in golden_util.load("288") if c["fn"] == "fast4" and c["note"] == "p=0.05")
    fl = golden_util.oracle_flags(case)
    a = oracle.decode_batch(case["H"], case["syndromes"], case["prior"], case["max_iter"], flags=fl)
    b = oracle.decode_batch(case["H"], case["syndromes"], case["prior"], case["max_iter"],
                            flags=fl | oracle.FLAG_LIBM_MATH)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    rel = np.abs(a[3] - b[3]) / np.maximum(np.abs(a[3]), 1e-300)
    assert 0 < rel.max() < 1e-3


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
    assert golden_util.same_bits(llr, case["llr"]).all()
