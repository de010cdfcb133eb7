"""CPU: the two restatements of numpy's float64 tanh / arctanh kernels -- oracle/np_math.h (test
infrastructure) and the host build of qldpc_amd/csrc/qbp_math.hpp (the functions the HIP kernels run) --
return numpy's BITS.

* against committed known-answer vectors produced by np.tanh / np.arctanh in the build container
  (tests/golden/np_math.npz, make_golden_np_math.py), on every host;
* against the live ufuncs on fresh inputs where this host's numpy dispatches to the same kernels
  (AVX512_SKX; elsewhere numpy falls back to libm and is itself a different function);
* the reciprocal step of the arctanh kernel (VRCP14PD rounded to 4 mantissa bits), tabulated by
  tools/extract_np_svml.py: device table == oracle thresholds, exhaustively over the 16 mantissa bits that
  matter and a range of exponents;
* accuracy against mpmath, for the record (these are numpy's functions: their error is the reference's).
"""
import ctypes as C
import os
import subprocess

import mpmath as mp
import numpy as np
import pytest

import golden_util
from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "np_math.npz"))


def _shim():
    src = os.path.join(HERE, "_shim", "math_host_shim.cpp")
    so = os.path.join(HERE, "_shim", "libmathshim_np.so")
    deps = [src] + [os.path.join(HERE, "..", "qldpc_amd", "csrc", f) for f in ("qbp_math.hpp", "qbp_np_tables.hpp")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-mfma", "-ffp-contract=off",
                               "-o", so, src])
    return C.CDLL(so)


def _call(fn, x):
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    fn(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_int64(x.size))
    return y


IMPLS = {
    "oracle": (lambda x: _call(oracle.lib().oracle_np_tanh, x), lambda y: _call(oracle.lib().oracle_np_arctanh, y)),
    # the kernels' forms: tanh(q / 2) of q = 2 x (exact doubling), and 2 * arctanh(y)
    "device-host-build": (lambda x: _call(_shim().shim_np_tanh_half, 2.0 * x),
                          lambda y: 0.5 * _call(_shim().shim_np_arctanh_x2, y)),
}


@pytest.mark.parametrize("impl", list(IMPLS))
def test_known_answer_vectors(impl):
    tanh_f, atanh_f = IMPLS[impl]
    x, ref = GOLD["tanh_x"], GOLD["tanh_y"]
    if impl != "oracle":
        # (the doubling overflows / is inexact only at the extremes of the range; a NaN message reaches the
        # kernels' np_tanh_half through tanh_half_msg<1>, which returns the NaN itself: qbp_math.hpp)
        keep = np.isinf(x) | ((np.abs(x) < 8e307) & ((np.abs(x) > 1e-300) | (x == 0)))
        x, ref = x[keep], ref[keep]
    with np.errstate(all="ignore"):
        got = tanh_f(x)
    assert golden_util.same_bits(got, ref).all(), f"{impl}: np.tanh bits differ on {int((~golden_util.same_bits(got, ref)).sum())}"
    y, ref = GOLD["atanh_x"], GOLD["atanh_y"]
    if impl != "oracle":          # (0.5 * (2 z) is z unless 2 z overflows: never here)
        keep = (np.abs(ref) > 1e-300) | (ref == 0) | ~np.isfinite(ref)
        y, ref = y[keep], ref[keep]
    with np.errstate(all="ignore"):
        got = atanh_f(y)
    assert golden_util.same_bits(got, ref).all(), f"{impl}: np.arctanh bits differ"


def _numpy_uses_svml():
    from numpy._core._multiarray_umath import __cpu_features__ as feat
    return bool(feat.get("AVX512_SKX"))


@pytest.mark.skipif(not _numpy_uses_svml(), reason="this host's numpy falls back to libm for tanh / arctanh")
@pytest.mark.parametrize("impl", list(IMPLS))
def test_against_live_numpy(impl):
    tanh_f, atanh_f = IMPLS[impl]
    rng = np.random.default_rng(int.from_bytes(os.urandom(4), "little"))
    x = np.concatenate([rng.normal(size=200000) * s for s in (1e-8, 0.05, 0.3, 1, 4, 20)])
    assert golden_util.same_bits(tanh_f(x), np.tanh(x)).all()
    y = np.clip(np.concatenate([rng.uniform(-1, 1, 400000), np.tanh(rng.normal(size=400000) * 4),
                                rng.choice([-1, 1], 200000) * (1 - 10.0 ** rng.uniform(-7, 0, 200000))]),
                -0.9999999, 0.9999999)
    assert golden_util.same_bits(atanh_f(y), np.arctanh(y)).all()


def test_check_message_is_the_references_expression():
    """qbp_math.hpp: check_message -- clip the magnitude, arctanh of the magnitude, sign bit (sign of x XOR
    syndrome bit) set at the end -- against the reference's own expression
    2.0 * np.arctanh(np.clip(x * syndrome_sign, -0.9999999, 0.9999999)) evaluated by numpy's goldens' rule:
    here through the oracle's np_arctanh, which the known-answer test pins to numpy."""
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.uniform(-1.5, 1.5, 200000), rng.normal(size=100000) * 1e-6, np.tanh(rng.normal(size=200000) * 4),
                        rng.choice([-1, 1], 50000) * (1 - 10.0 ** rng.uniform(-9, 0, 50000)),
                        np.array([0.0, -0.0, 1.0, -1.0, 0.9999999, -0.9999999, 1e-310, -1e-310, 5.0, -7.0, np.inf, -np.inf])])
    sbit = rng.integers(0, 2, len(x)).astype(np.uint8)
    want = 2.0 * _call(oracle.lib().oracle_np_arctanh, np.clip(x * (1.0 - 2.0 * sbit), -0.9999999, 0.9999999))
    got = np.empty_like(x)
    for variant in (0, 1):
        xs, sb = x, sbit
        if variant == 1:
            xs = np.concatenate([x, [np.nan]]); sb = np.concatenate([sbit, [1]]).astype(np.uint8)
            w = np.concatenate([want, [np.nan]])
        else:
            w = want
        got = np.empty_like(xs)
        _shim().shim_check_message(xs.ctypes.data_as(C.c_void_p), sb.ctypes.data_as(C.c_void_p),
                                   got.ctypes.data_as(C.c_void_p), C.c_long(len(xs)), C.c_int(variant))
        assert golden_util.same_bits(got, w).all(), variant


def test_reciprocal_table_exhaustive():
    """Device LUT form (qbp_math.hpp: np_rcp14_hi) == the threshold count of the oracle (np_math.h:
    np_rcp14_r4), for every value of the 16 mantissa bits that matter, other mantissa bits random, exponents
    of 1 + a (0) and 1 - a (down to 2^-24)."""
    import re
    txt = open(os.path.join(HERE, "..", "oracle", "np_svml_tables.h")).read()
    thr = [int(v, 16) for v in re.search(r"NP_RCP14_THR16\[16\] = \{([^}]*)\}", txt).group(1).split(",")]
    assert len(thr) == 16 and thr == sorted(thr)
    rng = np.random.default_rng(3)
    m16 = np.arange(65536, dtype=np.uint32)
    shim = _shim()
    for e in (0, -1, -2, -7, -24):
        v_hi = ((np.uint32(1023 + e) << np.uint32(20)) | (m16 << np.uint32(4)) |
                rng.integers(0, 16, 65536).astype(np.uint32)).astype(np.uint32)
        r_hi = np.empty_like(v_hi)
        shim.shim_np_rcp14_hi(v_hi.ctypes.data_as(C.c_void_p), r_hi.ctypes.data_as(C.c_void_p), C.c_long(v_hi.size))
        k = np.searchsorted(np.array(thr), m16, side="right").astype(np.int64)
        want = (0x3ff00000 - (k << 16) - (e << 20)).astype(np.uint32)
        assert np.array_equal(r_hi, want), f"exponent {e}"


def test_accuracy_for_the_record():
    mp.mp.prec = 120
    rng = np.random.default_rng(8)

    def worst(xs, got, fn):
        w = 0.0
        for x, g in zip(xs, got):
            ex = fn(mp.mpf(float(x)))
            if ex == 0:
                continue
            ulp = mp.mpf(2) ** (mp.floor(mp.log(abs(ex), 2)) - 52)
            w = max(w, float(abs(mp.mpf(float(g)) - ex) / ulp))
        return w
    xs = np.concatenate([rng.uniform(-20, 20, 3000), rng.uniform(-1, 1, 3000), rng.normal(size=1000) * 1e-3])
    wt = worst(xs, IMPLS["oracle"][0](xs), mp.tanh)
    ys = np.clip(np.concatenate([rng.uniform(-1, 1, 3000), np.tanh(rng.normal(size=3000) * 4)]), -0.9999999, 0.9999999)
    wa = worst(ys, IMPLS["oracle"][1](ys), mp.atanh)
    print(f"numpy's kernels against mpmath: tanh worst {wt:.2f} ulp, arctanh worst {wa:.2f} ulp")
    assert wt < 2.0 and wa < 1.0
