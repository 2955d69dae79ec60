"""Fits and re-measures the polynomial of csrc/common.h gelu_erf_f4 (CPU only: numpy + scipy).
erf(x / sqrt 2) = z Q(z^2 - 1), z = x sqrt 2 / L clamped to [-sqrt 2, sqrt 2]; Q of degree n - 1, iteratively re-weighted least squares on Chebyshev nodes
towards the minimax ABSOLUTE error of the GELU value 0.5 x (1 + erf(x / sqrt 2)); scaled so that the end value is exactly 1.  Prints the table and the
error of the fp32 fused-multiply-add evaluation over several ranges.    python tools/gelu_poly_fit.py [--L 4.5] [--n 10]"""
import argparse

import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=float, default=4.5)
    ap.add_argument("--n", type=int, default=10)
    a = ap.parse_args()
    L, n = a.L, a.n
    xs = np.cos(np.pi * (np.arange(8000) + 0.5) / 8000) * L
    xs = xs[xs > 0]
    z = xs * np.sqrt(2) / L
    sm = z * z - 1
    g = erf(xs / np.sqrt(2)) / z
    w = xs * np.maximum(xs, 0.5)
    for _ in range(300):
        c = C.chebfit(sm, g, n - 1, w=w)
        e = (C.chebval(sm, c) - g) * z * xs * 0.5
        w = w * (1 + 1.5 * np.abs(e) / np.abs(e).max())
        w /= w.max()
    coef = C.cheb2poly(c)
    coef = coef / (np.sqrt(2) * np.polyval(coef[::-1], 1.0))
    c32 = coef.astype(np.float32)

    def fma32(p, q, r):
        return (p.astype(np.float64) * q.astype(np.float64) + r.astype(np.float64)).astype(np.float32)

    def ev(x):
        x = x.astype(np.float32)
        zc = np.clip((x * np.float32(np.sqrt(2) / L)).astype(np.float32), np.float32(-np.sqrt(2)), np.float32(np.sqrt(2)))
        s = fma32(zc, zc, np.full_like(zc, -1))
        acc = np.full_like(s, c32[-1])
        for cc in c32[-2::-1]:
            acc = fma32(acc, s, np.full_like(s, cc))
        hx = (np.float32(0.5) * x).astype(np.float32)
        return fma32(hx, (zc * acc).astype(np.float32), hx)

    print("fit error (exact arithmetic): %.3e" % np.abs(e).max())
    for lo, hi in ((-4.5, 4.5), (-12, 12), (-200, 200), (-65504, 65504)):
        x = np.linspace(lo, hi, 2000001)
        xe = x.astype(np.float32).astype(np.float64)
        err = np.abs(ev(x) - 0.5 * xe * (1 + erf(xe / np.sqrt(2))))
        print(f"[{lo}, {hi}]: max abs error of the fp32 evaluation {err.max():.3e} at x = {x[err.argmax()]:.3f}")
    print("constexpr float kGeluErfC[%d] = {%s};" % (n, ", ".join(f"{v:.9e}f" for v in c32)))


if __name__ == "__main__":
    main()
