"""Column sums of heavy columns (>= 8 checks per variable).

The reference's loop form sums a gathered column with ``np.sum`` (decoding/beliefPropagation.py:68:
numpy's pairwise order from 8 terms on), its dense form accumulates row by row (:129).  The oracle
restates numpy's order (``np_pairwise_sum``, checked here bit for bit against numpy itself) behind
``FLAG_PAIRWISE_COLSUM``; the device does the same behind ``QBP_FLAG_PAIRWISE_COLSUM``, which
``performBeliefPropagation`` (the loop-form entry point) sets.  tests/golden/bp_heavy.npz holds
reference outputs of both forms on a 160 x 56 matrix with column weights up to 136.

Sum-product outputs of any two tanh/arctanh implementations differ in the last bits anyway, so the
order of a sum cannot be seen through them; min-sum has no transcendental, the oracle and the device
agree bit for bit there, and that is where the device's order is pinned against the oracle's."""
import numpy as np
import pytest

import golden_util
from oracle import oracle


def test_np_pairwise_sum_is_numpys():
    rng = np.random.default_rng(5)
    for L in list(range(0, 41)) + [63, 64, 65, 127, 128, 129, 136, 255, 256, 257, 300, 1000, 1031]:
        for _ in range(50):
            a = rng.normal(size=L) * 10.0 ** rng.integers(-3, 4)
            assert oracle.np_pairwise_sum(a) == float(np.sum(a)), L


def _flags(case):
    return oracle.FLAG_PAIRWISE_COLSUM if case["fn"] == "loop3" else 0


def test_oracle_heavy_goldens():
    n = 0
    for case in golden_util.load("heavy"):
        hard, conv, iters, llr = oracle.decode_batch(case["H"], case["syndromes"], case["prior"],
                                                     case["max_iter"], flags=_flags(case))
        golden_util.compare(case, hard, conv, iters, llr, "oracle")
        n += 1
    assert n == 4


def test_orders_differ_where_they_should():
    case = next(iter(golden_util.load("heavy")))
    kw = dict(variant=2, alpha=0.8, damping=0.7, clip_llr=25.0)
    # one iteration (later ones carry the difference into the light columns through the messages)
    a = oracle.decode_batch(case["H"], case["syndromes"], case["prior"], 1, **kw)
    b = oracle.decode_batch(case["H"], case["syndromes"], case["prior"], 1,
                            flags=oracle.FLAG_PAIRWISE_COLSUM, **kw)
    heavy = case["H"].sum(0) >= 8
    assert np.array_equal(a[3][:, ~heavy], b[3][:, ~heavy])        # light columns: same order
    assert (a[3][:, heavy] != b[3][:, heavy]).any()                # heavy ones: different bits


@pytest.mark.gpu
def test_device_heavy_goldens_and_order():
    from qldpc_amd import _lib, bp
    n = 0
    for case in golden_util.load("heavy"):
        dec = bp.decoder_for(case["H"])
        hard, conv, iters, llr = dec.decode(case["syndromes"], case["prior"], case["max_iter"],
                                            flags=golden_util.device_flags(case))
        golden_util.compare(case, hard, conv, iters, llr, "hip")
        n += 1
    assert n == 4
    # min-sum (no transcendental): device == oracle bit for bit, in either order
    kw = dict(alpha=0.8, damping=0.7, clip_llr=25.0)
    for fl_d, fl_o in ((0, 0), (_lib.FLAG_PAIRWISE_COLSUM, oracle.FLAG_PAIRWISE_COLSUM)):
        for mi in (1, 6, 30):
            d = dec.decode(case["syndromes"], case["prior"], mi, _lib.MIN_SUM, flags=fl_d, **kw)
            o = oracle.decode_batch(case["H"], case["syndromes"], case["prior"], mi, 2, flags=fl_o, **kw)
            # (weight-1 checks send +-inf in min-sum, so some values are inf or NaN: equal_nan)
            for x, y in zip(d, o):
                assert np.array_equal(x, y, equal_nan=x.dtype.kind == "f"), (fl_d, mi)


@pytest.mark.gpu
def test_loop_form_entry_point_uses_numpy_order(capsys):
    """performBeliefPropagation (loop form) on the heavy matrix == reference loop-form golden."""
    from qldpc_amd import bp
    case = next(c for c in golden_util.load("heavy") if c["fn"] == "loop3" and c["max_iter"] == 30)
    for i in range(4):
        hard, conv, llr = bp.performBeliefPropagation(case["H"], case["syndromes"][i], case["prior"],
                                                      verbose=False, maxIter=30)
        assert conv == bool(case["converged"][i]) and np.array_equal(hard, case["hard"][i])
        assert golden_util.same_bits(llr, case["llr"][i]).all()
