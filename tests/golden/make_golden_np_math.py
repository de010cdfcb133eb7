#!/usr/bin/env python3
"""Known-answer vectors of np.tanh / np.arctanh as numpy evaluates them in the build container
(numpy 2.2.6, AVX512_SKX dispatch -> its vendored SVML kernels): tests/golden/np_math.npz.

These two ufuncs are the only transcendental steps of the reference's BP (decoding/beliefPropagation.py:114,
:126); oracle/np_math.h and qldpc_amd/csrc/qbp_math.hpp restate them and must return these bits.
Run in the build container only:  python tests/golden/make_golden_np_math.py
"""
import os

import numpy as np
from numpy._core._multiarray_umath import __cpu_features__ as feat

assert feat.get("AVX512_SKX"), "numpy would use the libm fallback here: not the kernels the goldens come from"
rng = np.random.default_rng(20261005)
tx = np.concatenate(
    [rng.normal(size=5000) * s for s in (1e-300, 1e-12, 1e-3, 0.05, 0.2, 1.0, 3.0, 10.0, 30.0, 1e3)] +
    [rng.integers(0, 1 << 64, size=8000, dtype=np.uint64).view(np.float64),          # any bit pattern
     np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 0.125, 0.1875, 0.25, 0.375, 0.5, 0.75, 1.0, 1.5, 2, 3, 4, 6,
               8, 12, 16, 24, 23.999999999, 19.06, 700.0, 1e308, 1.7e308, 5e-324, -5e-324, 2.2250738585072014e-308])])
C0 = 0.9999999
ay = np.concatenate([
    rng.uniform(-1, 1, 15000), np.tanh(rng.normal(size=15000) * 5), rng.normal(size=10000) * 1e-5,
    rng.normal(size=5000) * 1e-200, rng.normal(size=2000) * 1e-310,
    rng.choice([-1, 1], 8000) * (1 - 10.0 ** rng.uniform(-7, 0, 8000)),
    # every rounding threshold of the kernel's reciprocal step, +-2 units of the 2^-16 mantissa grid (1 + a and 1 - a)
    np.array([s * (((t + d) / 65536.0)) for t in (0x040f, 0x0c98, 0x15b4, 0x1f70, 0x29e6, 0x3524, 0x4143, 0x4e60,
                                                    0x5c99, 0x6c16, 0x7d07, 0x8f9d, 0xa41a, 0xbad1, 0xd41c, 0xf082)
              for d in (-2, -1, 0, 1, 2) for s in (1.0, -1.0)]),
    np.array([0.0, -0.0, C0, -C0, 0.5, -0.5, 2.0 ** -1022, 5e-324, np.nan])])
ay = np.clip(ay, -C0, C0)          # (np.clip keeps NaN)
with np.errstate(all="ignore"):
    out = dict(tanh_x=tx, tanh_y=np.tanh(tx), atanh_x=ay, atanh_y=np.arctanh(ay),
               numpy_version=np.array(np.__version__), features=np.array(sorted(k for k, v in feat.items() if v)))
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "np_math.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes;", tx.size, "tanh and", ay.size, "arctanh vectors")
