#!/usr/bin/env python
"""Coefficients of the GELU the fp16 GEMM epilogues evaluate (csrc/vda_common.h, VDA_GELU_C0..6), CPU only:

    gelu(x) = max(x, 0) - |x| * P(|x|)^-16,     P(a) = c0 + c1 a + ... + c6 a^6   ( P^-16 ~ 0.5 erfc(a / sqrt 2) )

the form of Abramowitz-Stegun 7.1.28, refitted for the weight |x| by an L_p ladder (p = 2 ... 64, Nelder-Mead) started from the
published coefficients. Prints the coefficients and the error of an fp32 evaluation in the kernel's operation order."""
import math
import numpy as np
from scipy.optimize import minimize
from scipy.special import erf, erfc

ax = np.concatenate([np.linspace(0, 6, 6001), np.linspace(6, 12, 601)])
target = 0.5 * ax * erfc(ax / math.sqrt(2))
a = [.0705230784, .0422820123, .0092705272, .0001520143, .0002765672, .0000430638]
s = 2 ** (1 / 16)                                   # folds the 0.5
c = np.array([s] + [s * a[k] / math.sqrt(2) ** (k + 1) for k in range(6)])
resid = lambda c: ax * np.polyval(c[::-1], ax) ** -16.0 - target
print("A-S 7.1.28 as published: max abs %.3e" % np.abs(resid(c)).max())
with np.errstate(over="ignore"):
    for p in (2, 4, 8, 16, 32, 64):
        c = minimize(lambda c: np.sum((resid(c) * 1e6) ** p) ** (1.0 / p), c, method="Nelder-Mead",
                     options=dict(maxiter=20000, xatol=1e-14, fatol=1e-14, adaptive=True)).x
        print("L%-2d refit: max abs %.3e" % (p, np.abs(resid(c)).max()))
print("coefficients:", ", ".join("%.8e" % v for v in c))
x = np.linspace(-12, 12, 2000001).astype(np.float32)
ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / math.sqrt(2)))
co = [np.float32(v) for v in c]
axf = np.abs(x)
P = co[6] * axf + co[5]
for k in (4, 3, 2, 1, 0):
    P = (P * axf + co[k]).astype(np.float32)
r = (np.float32(1) / P).astype(np.float32)
for _ in range(4):
    r = (r * r).astype(np.float32)
e = np.abs((np.maximum(x, 0) - axf * r).astype(np.float64) - ref)
print("fp32 evaluation: max abs %.3e at x = %.3f; max relative where |gelu| > 1e-3: %.3e" % (e.max(), x[e.argmax()], (e / np.maximum(np.abs(ref), 1e-3)).max()))
