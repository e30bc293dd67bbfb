"""Coefficients of csrc/common.h: erf_fast (the per-message GELU of the edge kernels).
  |z| <= 1:  erf(z) = z + z p(z^2), p of degree 5
  |z| >  1:  erf(z) = sign(z) (1 - 2^(-q(|z|))), q of degree 6 fitted to -log2 erfc on [1, 4.2] (fp32 erf is 1 beyond)
Weighted least squares on Chebyshev nodes (weights = the factor that turns the fitted quantity's error into the absolute
error of erf), then the fp32 evaluation order of the kernel is replayed with numpy.float32 to report the real error."""
import numpy as np
from scipy.special import erf, erfc

f = np.float32


def cheb_nodes(a, b, n):
    k = np.arange(n)
    return 0.5 * (a + b) + 0.5 * (b - a) * np.cos((2 * k + 1) * np.pi / (2 * n))


def horner32(c, x):
    p = np.full_like(x, c[-1])
    for k in range(len(c) - 2, -1, -1):
        p = (p * x + c[k]).astype(f)
    return p


z = cheb_nodes(1e-6, 1.0, 4000)
A = np.vander(z * z, 6, increasing=True)
small = np.linalg.lstsq(A * z[:, None], (erf(z) / z - 1) * z, rcond=None)[0].astype(f)
zz = np.linspace(0, 1, 200001).astype(f)
r = (zz * horner32(small, (zz * zz).astype(f)) + zz).astype(f)
print("small:", [float(x) for x in small], "max |err| fp32 = %.3g" % np.abs(r - erf(zz.astype(np.float64))).max())

a = cheb_nodes(1.0, 4.2, 4000)
A = np.vander(a, 7, increasing=True)
w = erfc(a)
# the kernel evaluates 1 - exp2(-q2(a)) with v_exp_f32: log2(e) is folded into the coefficients
large = (np.linalg.lstsq(A * w[:, None], -np.log(erfc(a)) * w, rcond=None)[0] * np.log2(np.e)).astype(f)
aa = np.linspace(1, 8, 200001).astype(f)
q = horner32(large, np.minimum(aa, f(4.2)))
r = (f(1) - np.exp2(-q).astype(f)).astype(f)
print("large:", [float(x) for x in large], "max |err| fp32 = %.3g" % np.abs(r - erf(aa.astype(np.float64))).max())
