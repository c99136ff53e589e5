"""Minimax fit of the three coefficients of  gelu(u) ~= u * sigmoid(u * (c0 + c1 u^2 + c2 u^4))  used by gelu_fast /
gelu_grad_fast in simxns_amd/csrc/common.h, against the exact erf form; prints the max errors of value and derivative."""
import warnings

import numpy as np
from scipy.optimize import least_squares
from scipy.special import erf

warnings.filterwarnings("ignore")
Phi = lambda x: 0.5 * (1 + erf(x / np.sqrt(2)))
phi = lambda x: np.exp(-x * x / 2) / np.sqrt(2 * np.pi)
xs = np.linspace(-7, 7, 28001)
ref = xs * Phi(xs)


def model(c, x):
    s = x * x
    u = np.clip(x * (c[0] + s * (c[1] + s * c[2])), -80, 80)
    return x / (1 + np.exp(-u))


def grad(c, x):
    s = x * x
    r = 1 / (1 + np.exp(-np.clip(x * (c[0] + s * (c[1] + s * c[2])), -80, 80)))
    return r * (1 + x * (1 - r) * (c[0] + s * (3 * c[1] + s * 5 * c[2])))


c, w = np.array([1.5957691, 0.0713548, 0.0]), np.ones_like(xs)
for _ in range(80):
    c = least_squares(lambda c: (model(c, xs) - ref) * w, c).x
    e = np.abs(model(c, xs) - ref)
    w *= 1 + 2 * e / e.max()
    w /= w.mean()
print("c0, c1, c2 = %.10f, %.10f, %.12f" % tuple(c))
print("max |gelu error|  %.2e" % np.abs(model(c, xs) - ref).max())
print("max |gelu' error| %.2e" % np.abs(grad(c, xs) - (Phi(xs) + xs * phi(xs))).max())
