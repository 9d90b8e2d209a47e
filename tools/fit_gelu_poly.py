#!/usr/bin/env python
"""Minimax (Lawson-weighted least squares) odd polynomials for the bf16 GEMM epilogues' GELU (csrc/common.h: gelu_poly*):
    Phi(x)   - 0.5 ~= z P(z^2),  z = clamp(x / 4.0, -1, 1),  9 coefficients
    gelu'(x) - 0.5 ~= z Q(z^2),  z = clamp(x / 4.5, -1, 1), 10 coefficients
and their error INCLUDING fp32 Horner evaluation.  The transcendental form they replace (v_exp_f32 + v_rcp_f32 + 14 VALU per
element) made the fc1 / fc2-backward epilogues VALU-bound; these are 9-10 packed FMAs.  python tools/fit_gelu_poly.py"""
import numpy as np
from scipy.special import erf


def Phi(x):
    return 0.5 * (1 + erf(x / np.sqrt(2)))


def phi(x):
    return np.exp(-x * x / 2) / np.sqrt(2 * np.pi)


def fit(f, A, deg):
    n = 4000
    t = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    z = (t + 1) / 2
    V = np.stack([z * (z * z) ** k for k in range(deg)], 1)
    y = f(z * A)
    w = np.ones(n)
    for _ in range(200):
        c = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)[0]
        e = np.abs(V @ c - y)
        w = w * (e / e.max() + 1e-3) ** 0.5
        w /= w.max()
    return c


def horner32(c, z):
    z = z.astype(np.float32)
    s = z * z
    acc = np.full_like(z, np.float32(c[-1]))
    for k in range(len(c) - 2, -1, -1):
        acc = acc * s + np.float32(c[k])
    return z * acc


for name, f, A, deg in (("PHI", lambda x: Phi(x) - 0.5, 4.0, 9), ("DGELU", lambda x: Phi(x) - 0.5 + x * phi(x), 4.5, 10)):
    c = fit(f, A, deg)
    x = np.linspace(-8, 8, 400001)
    z = np.clip(x.astype(np.float32) * np.float32(1.0 / A), -1, 1)
    approx = horner32(c, z).astype(np.float64)
    err = np.abs(approx - f(x))
    print(f"// {name}: A = {A}, {deg} coefficients, max |err| on [-8, 8] = {err.max():.2e} (inside [-A, A]: {err[np.abs(x) <= A].max():.2e}), value at z = 1: {0.5 + horner32(c, np.ones(1))[0]:.7f}")
    print("constexpr float GELU_%s_C[%d] = {%s};" % (name, deg, ", ".join("%.9ef" % v for v in c)))
    if name == "PHI":
        g = x * (0.5 + approx)
        print(f"//   gelu(x) = x (0.5 + z P): max |err| = {np.abs(g - x * Phi(x)).max():.2e}, on [-4, 4]: {np.abs(g - x * Phi(x))[np.abs(x) <= 4].max():.2e}")
