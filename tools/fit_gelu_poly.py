#!/usr/bin/env python
"""Generator of the erf-GELU polynomial of the bf16 pipeline (easynlp_amd/csrc/ezclip_common.h: kPhiK).

Phi(x) ~ 0.5 + xc R(t),  xc = clamp(x, -c, c),  t = 2 xc^2 / c^2 - 1,  R of degree 8, minimax on [-c, c] (Lawson iterations on a
dense grid) UNDER THE CONSTRAINT R(1) = 0.5 / c: at the clamp the polynomial is exactly 0 / 1, so that outside it
gelu(x) = x Phi(x) is x (1 +- 1e-7) for x > c and x (0 +- 1e-7) for x < -c instead of x (Phi(+-c) +- fit error), whose error
grew with |x| (round 3's unconstrained fit: gelu(-12) = -7.2e-5).  The price is the fit error at the end points:
1 - Phi(c) = 1.33e-5 at c = 4.2 instead of the unconstrained 7.5e-6.  Prints the float32 coefficients (monomials in t, Horner)
and the measured errors of a float32 evaluation in the kernel's operation order.
"""
import math

import numpy as np

C = 4.2
DEG = 8


def phi(x):
    return 0.5 * (1.0 + np.vectorize(math.erf)(x / math.sqrt(2.0)))


def fit(c=C, deg=DEG, n=20001, iters=200):
    x = np.linspace(1e-6, c, n)
    t = 2 * x * x / (c * c) - 1
    target = (phi(x) - 0.5) / x - 0.5 / c          # = (t - 1) S(t)
    # basis: (t - 1) T_k(t), k = 0..deg-1
    T = np.polynomial.chebyshev.chebvander(t, deg - 1) * (t - 1)[:, None]
    w = np.ones(n)
    wx = x                                         # the error that matters is on Phi: x * (R - target)
    coef = None
    for _ in range(iters):
        sw = np.sqrt(w) * wx
        coef, *_ = np.linalg.lstsq(T * sw[:, None], target * sw, rcond=None)
        e = np.abs((T @ coef - target) * wx)
        w = w * (e / e.max() + 1e-3)
        w /= w.sum()
    # monomial coefficients of R(t) = 0.5 / c + (t - 1) S(t)
    S = np.polynomial.chebyshev.cheb2poly(coef)
    R = np.polynomial.polynomial.polymul(np.array([-1.0, 1.0]), S)
    R[0] += 0.5 / c
    return R


def eval_f32(k, x, c=C):
    x = x.astype(np.float32)
    xc = np.clip(x, np.float32(-c), np.float32(c))
    ts = np.float32(2.0 / (c * c))
    t = (xc.astype(np.float64) * xc.astype(np.float64)).astype(np.float32)      # v_mul_f32
    t = (t.astype(np.float64) * ts + (-1.0)).astype(np.float32)                 # fma
    r = np.full_like(x, np.float32(k[-1]))
    for kk in k[-2::-1]:
        r = (r.astype(np.float64) * t + np.float32(kk)).astype(np.float32)      # fma
    p = (xc.astype(np.float64) * r + 0.5).astype(np.float32)
    return p, (x.astype(np.float64) * p).astype(np.float32)


if __name__ == "__main__":
    R = fit()
    k = R.astype(np.float32)
    print("constexpr float kPhiK[%d] = {%s};" % (len(k), ", ".join("%.17gf" % float(v) for v in k)))
    x = np.linspace(-C, C, 400001)
    p, g = eval_f32(k, x)
    print("max |Phi error| on [-c, c]: %.3e" % np.abs(p - phi(x)).max())
    print("max |gelu error| on [-c, c]: %.3e" % np.abs(g - x * phi(x)).max())
    xo = np.concatenate([np.linspace(-60, -C, 20001), np.linspace(C, 60, 20001)])
    p, g = eval_f32(k, xo)
    print("outside the clamp: max |gelu error| %.3e, max |gelu error| / |x| %.3e" % (np.abs(g - xo * phi(xo)).max(),
                                                                                  (np.abs(g - xo * phi(xo)) / np.abs(xo)).max()))
    for v in (-50.0, -12.0, -5.0, 5.0, 12.0, 50.0):
        print("  gelu(%6.1f) = %.9g (exact %.9g)" % (v, eval_f32(k, np.array([v]))[1][0], v * phi(np.array([v]))[0]))
