#!/usr/bin/env python3
"""The coefficients of fast_abs_atan2f (mpc_amd/csrc/mpc_model.h): atan(t) = t + t s P(s), s = t^2, t in [0, 1], P of
degree 7, fitted for the smallest maximum RELATIVE error of atan(t) (Lawson's iteratively reweighted least squares on
Chebyshev nodes), then checked as the device evaluates it: every step rounded to FP32.  CPU only.

    python tools/fit_atan.py      ->  1.7e-8 relative in exact arithmetic, 1.5 ulp (8.9e-8) evaluated in FP32
"""
import numpy as np

N, DEG = 4000, 7
k = np.arange(N)
t = 0.5 * (1 - np.cos(np.pi * (k + 0.5) / N))
t = t[t > 1e-6]
s = t * t
target = (np.arctan(t) - t) / (t * s)
w = t * s / np.arctan(t)                        # d(relative error of atan) / d(P)
V = np.vander(s, DEG + 1, increasing=True)
lw = np.ones_like(s)
for _ in range(200):
    c = np.linalg.lstsq(V * (w * lw)[:, None], target * w * lw, rcond=None)[0]
    e = np.abs((V @ c - target) * w)
    lw = lw * (e / e.max() + 1e-3)
    lw /= lw.max()
print("max relative error, exact arithmetic: %.3g" % e.max())
c32 = c.astype(np.float32)
f32 = lambda x: np.asarray(x, dtype=np.float32)
tt = np.linspace(0, 1, 2000001).astype(np.float32)
ss = f32(tt * tt)
p = np.full_like(tt, c32[DEG])
for j in range(DEG - 1, -1, -1):
    p = f32(p.astype(np.float64) * ss.astype(np.float64) + np.float64(c32[j]))       # one FP32 fma per step
a = f32(tt.astype(np.float64) * f32(ss * p).astype(np.float64) + tt.astype(np.float64))
ref = np.arctan(tt.astype(np.float64))
rel = np.abs(a - ref)[1:] / ref[1:]
print("max relative error evaluated in FP32: %.3g = %.2f ulp" % (rel.max(), rel.max() / 2.0 ** -24))
print("coefficients, constant term first: " + ", ".join("%.9ef" % v for v in c32))
