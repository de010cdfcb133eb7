"""GPU: ulp error of the device FP64 tanh(q/2) and 2*atanh(y) against mpmath."""
import mpmath as mp
import numpy as np
import pytest

from qldpc_amd import bp, codes

pytestmark = pytest.mark.gpu


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


def test_device_math_ulp():
    dec = bp.decoder_for(codes.load_code("steane").Hx)
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.uniform(-40, 40, 3000), rng.uniform(-2, 2, 3000),
                         rng.choice([-1, 1], 1000) * 10.0 ** rng.uniform(-300, 1.7, 1000),
                         [0.0, 38.2, 40.0, 1e300, np.inf, -np.inf]])
    got = dec.debug_math(0, xs)
    fin = np.isfinite(xs)
    w = _ulps(got[fin], xs[fin], lambda v: mp.tanh(v / 2))
    assert got[-2] == 1.0 and got[-1] == -1.0
    print(f"device tanh_half worst {w:.3f} ulp")
    assert w <= 2.5
    C0 = 0.9999999
    ys = np.clip(np.concatenate([rng.uniform(-C0, C0, 4000),
                                 rng.choice([-1, 1], 2000) * (1 - 10.0 ** rng.uniform(-7, 0, 2000)),
                                 rng.choice([-1, 1], 1000) * 10.0 ** rng.uniform(-300, 0, 1000),
                                 [0.0, C0, -C0]]), -C0, C0)
    got = dec.debug_math(1, ys)
    w = _ulps(got, ys, lambda v: 2 * mp.atanh(v))
    print(f"device atanh2 worst {w:.3f} ulp")
    assert w <= 2.0
