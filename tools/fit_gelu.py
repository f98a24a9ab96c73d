#!/usr/bin/env python3
"""Coefficients of gelu_erf2 (hyper-vla_amd/csrc/common.h).

    gelu(x) = max(x, 0) - |x| Phi(-|x|),   Phi(-a) = erfc(a / sqrt 2) / 2 ~ exp2(q(a)),  0 <= a <= 8

q is a degree-5 polynomial (degree 6 in rounds 4-5: 8.7e-8 instead of 4.7e-7): a weighted least-squares fit of log2 Phi(-a), re-weighted towards the minimax solution of the
error that matters, |a Phi(-a) (2^(q - log2 Phi) - 1)| = the absolute error of the GELU value.  Prints the monomial
coefficients (Horner order of the kernel: highest first) and the error of the f32 evaluation the kernel performs, next to
the Abramowitz-Stegun 7.1.26 form it replaced.  CPU only (numpy + scipy).
"""
import numpy as np
from numpy.polynomial import Polynomial, chebyshev as C
from scipy.special import erfc

LIM, DEG = 8.0, 5          # degree 6 until round 5 (git history: the A/B switch was pruned in round 6)


def main():
    a = np.linspace(0.0, LIM, 400001)
    phi = 0.5 * erfc(a / np.sqrt(2.0))
    f = np.log2(phi)
    xs = 2.0 * a / LIM - 1.0
    w = a * phi + 1e-9
    for _ in range(60):
        c = C.chebfit(xs, f, DEG, w=w)
        d = C.chebval(xs, c) - f
        err = np.abs(a * phi * (2.0 ** d - 1.0))
        w = w * (1.0 + 2.0 * err / err.max())
    mono = Polynomial(C.cheb2poly(c))(Polynomial([-1.0, 2.0 / LIM])).coef
    print("max |error of a Phi(-a)| in exact arithmetic: %.3e" % err.max())
    print("coefficients, highest degree first (f32):")
    for m in mono[::-1]:
        print("  %.16g" % float(np.float32(m)))
    a32 = a.astype(np.float32)
    q = np.full_like(a32, np.float32(mono[-1]))
    for m in mono[-2::-1]:
        q = (q.astype(np.float64) * a32 + np.float32(m)).astype(np.float32)          # one rounding per fma
    h = (a32.astype(np.float64) * np.exp2(q.astype(np.float64)).astype(np.float32)).astype(np.float32)
    exact = a32.astype(np.float64) * 0.5 * erfc(a32.astype(np.float64) / np.sqrt(2.0))
    print("evaluated in f32: max |error| %.3e, q(%.0f) = %.2f" % (np.abs(h - exact).max(), LIM, q[-1]))
    t = 1.0 / (1.0 + 0.3275911 * a / np.sqrt(2.0))
    p = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429))))
    print("Abramowitz-Stegun 7.1.26 (exact arithmetic): max |error| %.3e" % np.abs(a * 0.5 * p * np.exp(-a * a / 2) - a * phi).max())


if __name__ == "__main__":
    main()
