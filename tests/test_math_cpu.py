"""CPU: ulp error of the FP64 tanh(q/2) / 2*atanh(y) used by the HIP kernels (host build of
qldpc_amd/csrc/qbp_math.hpp) against mpmath."""
import ctypes as C
import os
import subprocess

import mpmath as mp
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _build(tag, extra):
    src = os.path.join(HERE, "_shim", "math_host_shim.cpp")
    so = os.path.join(HERE, "_shim", f"libmathshim{tag}.so")
    hdr = os.path.join(HERE, "..", "qldpc_amd", "csrc", "qbp_math.hpp")
    if (not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src),
                                                             os.path.getmtime(hdr))):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-mfma",
                               "-ffp-contract=off", *extra, "-o", so, src])
    return C.CDLL(so)


# The device seeds its divisions with v_rcp_f64 (relative error up to 2^-24.4 = 4.6e-8, measured);
# the host build's seed is exact, so the suite also runs with the seed perturbed by +-5e-8.
@pytest.fixture(scope="module", params=["", "_p", "_m"])
def shim(request):
    extra = {"": [], "_p": ["-DQBP_TEST_SEED_ERR=5e-8"], "_m": ["-DQBP_TEST_SEED_ERR=-5e-8"]}
    return _build(request.param, extra[request.param])


def _call(fn, x):
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    fn(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_long(x.size))
    return y


def _ulp_err(got, exact_fn, xs):
    mp.mp.prec = 120
    worst, arg = 0.0, None
    for x, g in zip(xs, got):
        ex = exact_fn(mp.mpf(float(x)))
        if ex == 0:
            assert g == 0
            continue
        ulp = mp.mpf(2) ** (mp.floor(mp.log(abs(ex), 2)) - 52)
        err = float(abs(mp.mpf(float(g)) - ex) / ulp)
        if err > worst:
            worst, arg = err, float(x)
    return worst, arg


def test_tanh_half_ulp(shim):
    rng = np.random.default_rng(1)
    xs = np.concatenate([
        rng.uniform(-40, 40, 4000), rng.uniform(-2, 2, 4000),
        rng.choice([-1, 1], 2000) * 10.0 ** rng.uniform(-300, 1.7, 2000),
        [0.0, -0.0, 1e-320, 38.0, 38.2, 40.0, 41.0, 1e300, -1e300, np.inf, -np.inf,
         np.log(2), 2 * np.log(2), 0.34657359, 0.6931471805599453, 4.59511985013459]])
    got = _call(shim.shim_tanh_half, xs)
    assert got[np.isinf(xs)].tolist() == np.sign(xs[np.isinf(xs)]).tolist()
    z = _call(shim.shim_tanh_half, np.array([0.0, -0.0]))
    assert z[0] == 0 and z[1] == 0 and np.signbit(z[1]) and not np.signbit(z[0])
    fin = np.isfinite(xs)
    worst, arg = _ulp_err(got[fin], lambda v: mp.tanh(v / 2), xs[fin])
    print(f"tanh_half worst {worst:.3f} ulp at {arg!r}")
    assert worst <= 2.5     # numerator, denominator and division round independently (glibc: 2 ulp)
    assert (np.abs(got) <= 1.0).all()
    # monotone saturation: exactly 1 for |q| >= 38.2 like a correctly rounded tanh
    assert got[np.abs(xs) >= 38.2].tolist() == np.sign(xs[np.abs(xs) >= 38.2]).tolist()


def test_atanh2_ulp(shim):
    rng = np.random.default_rng(2)
    C0 = 0.9999999
    xs = np.concatenate([
        rng.uniform(-C0, C0, 6000),
        rng.choice([-1, 1], 3000) * (1 - 10.0 ** rng.uniform(-7, 0, 3000)),
        rng.choice([-1, 1], 3000) * 10.0 ** rng.uniform(-300, 0, 3000) * C0,
        [0.0, -0.0, C0, -C0, 0.1715728752538099, 0.17157287525381, 0.1715, 0.172, 0.5, 1 / 3,
         0.6568542494923802, 1e-320]])
    xs = np.clip(xs, -C0, C0)
    got = _call(shim.shim_atanh2, xs)
    z = _call(shim.shim_atanh2, np.array([0.0, -0.0]))
    assert z[0] == 0 and z[1] == 0 and np.signbit(z[1]) and not np.signbit(z[0])
    worst, arg = _ulp_err(got, lambda v: 2 * mp.atanh(v), xs)
    print(f"atanh2 worst {worst:.3f} ulp at {arg!r}")
    assert worst <= 2.0


def test_div_nr(shim):
    rng = np.random.default_rng(3)
    a = rng.uniform(-2, 2, 400000)
    b = np.concatenate([rng.uniform(1e-7, 4, 200000), 10.0 ** rng.uniform(-15, 0, 200000)])
    y = np.empty_like(a)
    shim.shim_div(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                  y.ctypes.data_as(C.c_void_p), C.c_long(a.size))
    exact = a / b
    ulps = np.abs(y - exact) / np.spacing(np.abs(exact))
    assert ulps.max() <= 1.0
    assert np.mean(y == exact) >= 0.99999
    print(f"div_nr: {np.mean(y == exact) * 100:.4f}% correctly rounded, worst {ulps.max():.1f} ulp")
