#!/usr/bin/env python3
"""Derive the polynomial coefficients used by qldpc_amd/csrc/qbp_math.hpp with mpmath.

  expm1(r) = r + r^2 * P(r)          on |r| <= ln2/2      (tanh(q/2) via expm1(-|q|))
  2*atanh(s) = 2 s + s * z * L(z)    z = s^2, |s| <= 0.1716 (= 3 - 2 sqrt 2)

Near-minimax fits from mpmath.chebyfit; prints C initialisers and the observed max error.
"""
import mpmath as mp

mp.mp.prec = 200


def fit(f, lo, hi, n):
    c = mp.chebyfit(f, [lo, hi], n)     # highest power first
    return [float(x) for x in c][::-1]  # lowest power first, rounded to double


def horner(c, x):
    acc = mp.mpf(0)
    for k in reversed(c):
        acc = acc * x + mp.mpf(k)
    return acc


def report(name, c):
    print(f"// {name}")
    for i, v in enumerate(c):
        print(f"    {v.hex()},  // {v!r}  x^{i}")


hl = mp.log(2) / 2 * mp.mpf("1.02")
for n in (11, 12, 13):
    P = lambda r: (mp.expm1(r) - r) / (r * r) if r != 0 else mp.mpf("0.5")
    c = fit(P, -hl, hl, n)
    worst = 0
    for i in range(-2000, 2001):
        r = hl * i / 2000
        if r == 0:
            continue
        approx = r + r * r * horner(c, r)
        worst = max(worst, abs(approx / mp.expm1(r) - 1))
    print(f"expm1 P with {n} coeffs: max rel err {mp.nstr(worst, 3)} = {mp.nstr(worst / mp.mpf(2) ** -53, 3)} ulp-ish")
    if n == 12:
        report("EXPM1_P", c)

# tanh(h) = h * T(h^2) on |h| <= ln2/4: the form qbp_math.hpp uses (the expm1 fit above belongs to
# an earlier, 4-operations-longer tanh and is kept for reference)
hq = mp.log(2) / 4 * mp.mpf("1.02")
for n in (6, 7, 8):
    Tf = lambda z: (mp.tanh(mp.sqrt(z)) / mp.sqrt(z)) if z != 0 else mp.mpf(1)
    c = fit(Tf, 0, hq * hq, n)
    worst = 0
    for i in range(1, 2001):
        h = hq * i / 2000
        worst = max(worst, abs(h * horner(c, h * h) / mp.tanh(h) - 1))
    print(f"tanh T with {n} coeffs: max rel err {mp.nstr(worst, 3)} = {mp.nstr(worst / mp.mpf(2) ** -53, 3)} ulp-ish")
    if n == 7:
        report("TANH_T", c)

smax = (3 - 2 * mp.sqrt(2)) * mp.mpf("1.02")
for n in (6, 7, 8):
    L = lambda z: ((2 * mp.atanh(mp.sqrt(z)) - 2 * mp.sqrt(z)) / (mp.sqrt(z) * z)) if z != 0 else mp.mpf(2) / 3
    c = fit(L, 0, smax * smax, n)
    worst = 0
    for i in range(1, 2001):
        s = smax * i / 2000
        z = s * s
        approx = 2 * s + s * z * horner(c, z)
        worst = max(worst, abs(approx / (2 * mp.atanh(s)) - 1))
    print(f"atanh L with {n} coeffs: max rel err {mp.nstr(worst, 3)} = {mp.nstr(worst / mp.mpf(2) ** -53, 3)} ulp-ish")
    if n == 7:
        report("ATANH_L", c)
