"""GPU: the FP64 tanh(q/2) and 2*atanh(y) the kernels run (qldpc_amd/csrc/qbp_math.hpp) return numpy's bits
on the committed known-answer vectors of np.tanh / np.arctanh (tests/golden/np_math.npz); the round-1/2
forms kept for A/B builds (QBP_MATH_FAST) are still within their ulp bounds against mpmath."""
import os

import mpmath as mp
import numpy as np
import pytest

import golden_util
from qldpc_amd import bp, codes

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "np_math.npz"))


def _ulps(got, xs, fn):
    mp.mp.prec = 120
    worst = 0.0
    for x, g in zip(xs, got):
        ex = fn(mp.mpf(float(x)))
        if ex == 0:
            assert g == 0
            continue
        ulp = mp.mpf(2) ** (mp.floor(mp.log(abs(ex), 2)) - 52)
        worst = max(worst, float(abs(mp.mpf(float(g)) - ex) / ulp))
    return worst


def test_device_math_returns_numpys_bits():
    dec = bp.decoder_for(codes.load_code("steane").Hx)
    x, ref = GOLD["tanh_x"], GOLD["tanh_y"]
    # the kernels evaluate tanh(q / 2): q = 2 x (exact away from the ends of the range; a NaN message goes
    # through tanh_half_msg<1>, not through this entry)
    keep = np.isinf(x) | ((np.abs(x) < 8e307) & ((np.abs(x) > 1e-300) | (x == 0)))
    got = dec.debug_math(0, 2.0 * x[keep])
    assert golden_util.same_bits(got, ref[keep]).all()
    y, ref = GOLD["atanh_x"], GOLD["atanh_y"]
    keep = ~np.isnan(y)
    got = dec.debug_math(1, y[keep])                      # 2 * arctanh(y)
    assert golden_util.same_bits(got, 2.0 * ref[keep]).all()
    print(f"device: {int(keep.sum())} arctanh and {len(x)} tanh known answers of numpy reproduced bit for bit")


def test_fast_forms_ulp():
    dec = bp.decoder_for(codes.load_code("steane").Hx)
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.uniform(-40, 40, 3000), rng.uniform(-2, 2, 3000),
                         rng.choice([-1, 1], 1000) * 10.0 ** rng.uniform(-300, 1.7, 1000),
                         [0.0, 38.2, 40.0, 1e300, np.inf, -np.inf]])
    got = dec.debug_math(4, xs)
    fin = np.isfinite(xs)
    w = _ulps(got[fin], xs[fin], lambda v: mp.tanh(v / 2))
    assert got[-2] == 1.0 and got[-1] == -1.0
    print(f"device tanh_half (fast form) worst {w:.3f} ulp")
    assert w <= 2.5
    C0 = 0.9999999
    ys = np.clip(np.concatenate([rng.uniform(-C0, C0, 4000),
                                 rng.choice([-1, 1], 2000) * (1 - 10.0 ** rng.uniform(-7, 0, 2000)),
                                 rng.choice([-1, 1], 1000) * 10.0 ** rng.uniform(-300, 0, 1000),
                                 [0.0, C0, -C0]]), -C0, C0)
    got = dec.debug_math(5, ys)
    w = _ulps(got, ys, lambda v: 2 * mp.atanh(v))
    print(f"device atanh2 (fast form) worst {w:.3f} ulp")
    assert w <= 2.0
