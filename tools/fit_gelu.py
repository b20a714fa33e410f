"""Fit and check the single-piece GELU of the step loop (csrc/sim_device.hpp: gelu_fast).

gelu(v) = max(v,0) - |v| * 2^P(|v|),  P(t) ~= log2(erfc(t/sqrt2)/2)  on [0,6], degree 8, weighted so that the
absolute error of the RESULT is minimised.  Prints the fp32 coefficients and the error (fp32 Horner with fused
multiply-adds, against fp64 x*Phi(x)), next to the error of torch's fp32 formula (0.5 v)(1 + erf(v/sqrt2))
evaluated with an exact erf.
"""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
from scipy.special import erf, erfc

T, DEG = 6.0, 8


def target(t):
    return np.log2(0.5 * erfc(t / np.sqrt(2)))


def fit(deg, iters=30):
    t = np.linspace(0, T, 20001)
    y = target(t)
    w0 = 0.5 * erfc(t / np.sqrt(2)) * np.maximum(t, 0.05)
    w = w0.copy()
    for _ in range(iters):  # iteratively re-weighted least squares -> near-minimax
        c = C.chebfit(2 * t / T - 1, y, deg, w=w)
        e = (C.chebval(2 * t / T - 1, c) - y) * w0
        w = w * (1 + 0.5 * np.abs(e) / np.abs(e).max())
    pc, u, mono, res = C.cheb2poly(c), np.array([-1.0, 2.0 / T]), np.array([1.0]), np.zeros(deg + 1)
    for ck in pc:
        res[:len(mono)] += ck * mono
        mono = P.polymul(mono, u)
    return res.astype(np.float32)


def gelu_poly(v, coef):
    t = np.abs(v)
    r = np.full_like(t, coef[-1])
    for k in range(len(coef) - 2, -1, -1):
        r = (r.astype(np.float64) * t + coef[k]).astype(np.float32)
    p = np.exp2(r.astype(np.float64)).astype(np.float32)
    return (np.maximum(v, 0) - t.astype(np.float64) * p).astype(np.float32)


def gelu_torch_formula(v):
    e = erf((v * np.float32(0.70710678118654752440)).astype(np.float32).astype(np.float64)).astype(np.float32)
    return ((v * np.float32(0.5)).astype(np.float32) * (np.float32(1) + e).astype(np.float32)).astype(np.float32)


if __name__ == "__main__":
    coef = fit(DEG)
    print("coefficients (low -> high):", ", ".join(f"{c:.9e}f" for c in coef))
    v = np.linspace(-8, 8, 1600001).astype(np.float32)
    ref = 0.5 * v.astype(np.float64) * (1 + erf(v.astype(np.float64) / np.sqrt(2)))
    for name, out in (("poly", gelu_poly(v, coef)), ("torch formula, exact erf", gelu_torch_formula(v))):
        err = np.abs(out - ref)
        for lo, hi in ((0, 0.5), (0.5, 1.5), (1.5, 3), (3, 8)):
            m = (np.abs(v) >= lo) & (np.abs(v) < hi)
            print(f"{name:26s} |v| in [{lo},{hi}): max abs err {err[m].max():.3e}")
    t = np.linspace(6, 200, 200001)
    r = np.polyval(coef[::-1].astype(np.float64), t)
    print("P monotone decreasing beyond 6:", bool(np.all(np.diff(r) < 0)), " P(6) =", r[0])
