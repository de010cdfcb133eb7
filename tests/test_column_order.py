"""CPU: the order in which the library's tables add up a column of messages (qbp_column_order, host-only)
reproduces numpy itself -- np.sum(R, axis=0) on a C-ordered and on a Fortran-ordered dense R -- and the
drop-in layer picks the order from the memory order of the caller's H exactly as the reference's numpy
code ends up doing (decoding/beliefPropagation.py:129; rework/decoding.py:61, :119, :173)."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

from qldpc_amd import _lib, bp, codes


def _order(H, col_order):
    rp, ci, m, n = bp.csr_from_H(H)
    E = int(rp[-1])
    cp = np.zeros(n + 1, np.int32)
    ce = np.zeros(max(E, 1), np.int32)
    rc = _lib.load().qbp_column_order(rp.ctypes.data, ci.ctypes.data, m, n, col_order, cp.ctypes.data, ce.ctypes.data)
    return rc, rp, ci, cp, ce


def _left_to_right(vals, cp, ce, n):
    out = np.zeros(n)
    for v in range(n):
        s, first = 0.0, True
        for q in range(cp[v], cp[v + 1]):
            s = vals[ce[q]] if first else s + vals[ce[q]]
            first = False
        out[v] = s
    return out


def _random_H(rng, m, n, w):
    H = np.zeros((m, n), np.int64)
    for v in range(n):
        H[rng.choice(m, rng.integers(0, w + 1), replace=False), v] = 1
    return H


@pytest.mark.parametrize("shape", ["72", "90", "108", "144", "288", (7, 20, 2), (50, 100, 3), (130, 260, 3),
                                   (300, 500, 3), (1000, 1500, 3)])
def test_tables_reproduce_numpy_sums_in_both_memory_orders(shape):
    rng = np.random.default_rng(11)
    H = codes.load_code(shape).Hx if isinstance(shape, str) else _random_H(rng, *shape)
    m, n = H.shape
    for col_order, layout in ((0, np.ascontiguousarray), (1, np.asfortranarray)):
        rc, rp, ci, cp, ce = _order(H, col_order)
        assert rc == 0
        rows = np.repeat(np.arange(m), np.diff(rp))
        for _ in range(10):
            vals = rng.normal(size=int(rp[-1])) * 10.0 ** rng.integers(-3, 3, size=int(rp[-1]))
            R = np.zeros((m, n))
            R[rows, ci] = vals
            ref = np.sum(layout(R), axis=0)
            assert np.array_equal(_left_to_right(vals, cp, ce, n), ref), (shape, col_order)


def test_balanced_association_is_refused_not_approximated():
    """Four entries in four different running sums: numpy adds them (a + b) + (c + d) -- no left-to-right
    order gives that, and the library says so (QBP_E_UNSUPPORTED) instead of returning other bits."""
    H = np.zeros((16, 2), np.int64)
    H[[0, 1, 2, 3], 0] = 1
    H[[0, 1], 1] = 1
    rc, *_ = _order(H, 1)
    assert rc == _lib.E_UNSUPPORTED
    assert _order(H, 0)[0] == 0


def test_flag_choice_follows_numpy_layout_rules():
    Hf = codes.load_code("288").Hx
    assert Hf.flags.f_contiguous and not Hf.flags.c_contiguous         # as the reference's code files
    Hc = np.ascontiguousarray(Hf)
    assert bp.dense_colsum_flags(Hc) == 0 and bp.dense_colsum_flags(Hc, damped=True) == 0
    assert bp.dense_colsum_flags(Hf) == _lib.FLAG_DENSE_F_COLSUM
    assert bp.dense_colsum_flags(Hf, damped=True) == _lib.FLAG_DENSE_F_COLSUM         # 144 x 288 x 8 >= 256 KiB
    H144 = codes.load_code("144").Hx
    assert bp.dense_colsum_flags(H144, damped=True) == _lib.FLAG_DENSE_F_COLSUM_ITER0
    assert bp.dense_colsum_flags(H144.T.copy().T) == _lib.FLAG_DENSE_F_COLSUM       # F-like strides
    assert bp.dense_colsum_flags(Hf.tolist()) == 0
    from scipy.sparse import csr_matrix
    assert bp.dense_colsum_flags(csr_matrix(Hf)) == 0


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="build container only: replays the reference's own numpy code")
def test_layout_rules_against_the_reference_itself():
    """The three rules of bp.dense_colsum_flags, observed on the reference's own functions: same inputs,
    H in C order vs Fortran order -> the LLR bits differ exactly where the rules say the order does."""
    spec = importlib.util.spec_from_file_location("ref_rework_decoding", os.path.join(REF, "rework", "decoding.py"))
    rework = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rework)
    from oracle import oracle
    rng = np.random.default_rng(5)
    for tag in ("144", "288"):
        code = codes.load_code(tag)
        p = 0.06
        prior = np.log((1 - p) / p) * rng.uniform(0.7, 1.3, code.n)          # non-uniform: iteration 0 matters
        for _ in range(6):
            s = ((rng.random(code.n) < p).astype(np.int64) @ code.Hx.T % 2).astype(np.int8)
            for H in (np.ascontiguousarray(code.Hx), np.asfortranarray(code.Hx)):
                for fn, name, kw in ((rework.performBeliefPropagationFast, "fast4", {}),
                                     (rework.performMinSum_Symmetric, "minsum", dict(alpha=0.8, damping=0.7, clip_llr=25.0)),
                                     (rework.performBeliefPropagation_Symmetric, "sym", dict(alpha=0.9, damping=0.8, clip_llr=20.0))):
                    h, c, llr, it = fn(H, s, prior, maxIter=30, **kw)
                    variant = {"fast4": 0, "minsum": 2, "sym": 1}[name]
                    o = oracle.decode_batch(H, s[None, :].astype(np.uint8), prior, 30, variant,
                                            flags=oracle.colsum_flags(name, H), **kw)
                    assert o[2][0] == it and bool(o[1][0]) == bool(c) and np.array_equal(o[0][0], h)
                    assert (o[3][0].view(np.uint64) == np.asarray(llr).view(np.uint64)).all(), (tag, name, H.flags.f_contiguous)
