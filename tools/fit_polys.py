#!/usr/bin/env python
"""Generate the polynomial coefficients used by pyvbmc_amd/csrc/fastmath.h.

High-precision (mpmath) Chebyshev-node interpolation -- within a small factor of
minimax -- of
    2^f           on f in [-1/2, 1/2]            (device exp2 after range reduction)
    atanh(s)/s    in s^2, |s| <= (sqrt(2)-1)/(sqrt(2)+1)  (device log: ln m = 2 atanh((m-1)/(m+1)))
    sin(pi x)/x, cos(pi x)  on |x| <= 1/4      (Box-Muller angle)
and print them as C hex-float initialisers together with the measured max relative
error in float64 Horner evaluation.
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60


def cheb_fit(fn, a, b, n):
    """degree-n interpolant at Chebyshev nodes on [a,b]; returns monomial coeffs in x."""
    nodes = [mp.cos(mp.pi * (2 * i + 1) / (2 * (n + 1))) for i in range(n + 1)]
    xs = [(a + b) / 2 + (b - a) / 2 * t for t in nodes]
    A = mp.matrix(n + 1, n + 1)
    y = mp.matrix(n + 1, 1)
    for i, x in enumerate(xs):
        for j in range(n + 1):
            A[i, j] = x**j
        y[i] = fn(x)
    c = mp.lu_solve(A, y)
    return [c[i] for i in range(n + 1)]


def horner64(coef, x):
    acc = np.full_like(x, float(coef[-1]))
    for a in coef[-2::-1]:
        acc = acc * x + float(a)
    return acc


def show(name, coef):
    print(f"// {name}")
    print("{" + ", ".join(float(c).hex() for c in coef) + "}")


if __name__ == "__main__":
    half = mp.mpf(1) / 2
    for n in (8, 9, 10, 11, 12):  # 8: the entropy kernels' (fastmath.h VBMC_ENT_EXP2_COEFFS, c0 = 1); 11: exp2_fast
        c = cheb_fit(lambda x: mp.mpf(2) ** x, -half, half, n)
        f = np.linspace(-0.5, 0.5, 400001)
        err = np.max(np.abs(horner64(c, f) / np.exp2(f) - 1))
        show(f"exp2 degree {n}: max rel err {err:.3e}", c)
    # log: ln(m) = 2 s (1 + s^2/3 + s^4/5 + ...) , s = (m-1)/(m+1), m in [sqrt(1/2), sqrt(2)]
    smax = (mp.sqrt(2) - 1) / (mp.sqrt(2) + 1)
    for n in (8, 9, 10):
        c = cheb_fit(lambda u: (mp.atanh(mp.sqrt(u)) / mp.sqrt(u)) if u > 0 else mp.mpf(1), mp.mpf(0), smax**2, n)
        m = np.linspace(2**-0.5, 2**0.5, 400001)
        s = (m - 1) / (m + 1)
        val = 2 * s * horner64(c, s * s)
        ref = np.array([float(mp.log(mp.mpf(float(x)))) for x in m[::400]])
        err = np.max(np.abs(val[::400] - ref))
        show(f"atanh(s)/s in u=s^2 degree {n}: max abs err of ln m {err:.3e}", c)
    q = mp.mpf(1) / 4
    for n in (6, 7):
        c = cheb_fit(lambda u: (mp.sin(mp.pi * mp.sqrt(u)) / mp.sqrt(u)) if u > 0 else mp.pi, mp.mpf(0), q * q, n)
        x = np.linspace(-0.25, 0.25, 200001)
        err = np.max(np.abs(x * horner64(c, x * x) - np.sin(np.pi * x)))
        show(f"sin(pi x)/x in u=x^2 degree {n}: max abs err {err:.3e}", c)
        c = cheb_fit(lambda u: mp.cos(mp.pi * mp.sqrt(u)), mp.mpf(0), q * q, n)
        err = np.max(np.abs(horner64(c, x * x) - np.cos(np.pi * x)))
        show(f"cos(pi x) in u=x^2 degree {n}: max abs err {err:.3e}", c)
