"""Coefficients of composite.hip's h(x) = erfc(x)/2 = 2^Q(x'), x' = x*sqrt(log2 e), x in [0, 5].
Weighted (Lawson) minimax fit of log2(erfc/2) so that the ABSOLUTE error of 2^Q is minimised;
the fp32 Horner/fma evaluation is emulated to report the error the kernel actually sees."""
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erfc

c = np.sqrt(np.log2(np.e))


def fit(n, xmax, iters=40):
    xs = np.cos(np.linspace(0, np.pi, 6001)) * xmax / 2 + xmax / 2
    g = np.log2(erfc(xs / c) / 2)
    h = erfc(xs / c) / 2
    lw = np.ones_like(xs)
    for _ in range(iters):
        cc = C.chebfit(2 * xs / xmax - 1, g, n, w=h * lw)
        err = np.abs(C.chebval(2 * xs / xmax - 1, cc) - g) * h
        lw = lw * (err / err.max() + 1e-3)
        lw /= lw.mean()
    P = np.polynomial.Polynomial(C.cheb2poly(cc))(np.polynomial.Polynomial([-1, 2 / xmax]))
    return P.coef


def horner_fma32(coef, x):
    x = x.astype(np.float32).astype(np.float64)
    acc = np.full_like(x, np.float64(np.float32(coef[-1])))
    for cf in coef[-2::-1]:
        acc = (acc * x + np.float64(np.float32(cf))).astype(np.float32).astype(np.float64)
    return np.exp2(acc).astype(np.float32)


if __name__ == "__main__":
    xmax = 5.0 * c
    for n in (6, 7, 8):
        coef = fit(n, xmax)
        x = np.linspace(0, xmax, 400001)
        e32 = np.abs(horner_fma32(coef, x).astype(np.float64) - erfc(x / c) / 2)
        print(n, "max abs err (fp32 eval) %.3e" % e32.max(), " h(0) =", horner_fma32(coef, np.zeros(1))[0])
        print("   ", ", ".join("%.9ef" % np.float32(v) for v in coef))
