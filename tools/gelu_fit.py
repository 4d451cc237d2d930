"""Fit and check of the bf16-path GELU of csrc/cwlt_gelu.h (CPU, numpy/scipy).

    Phi(-|x|) = exp(-x^2 / 2) * Q(|x|),   Q of degree 5 with Q(0) = 1/2

Minimises max over x >= 0 of max(|gelu error|, |gelu' error|) (the density is exact, so both errors are
|e Q - Phi(-x)| times x resp. 1), then evaluates the device's instruction sequence in emulated f32 arithmetic on
[-14, 14].  Prints the coefficients that cwlt_gelu.h holds and the errors quoted there."""
import numpy as np
from scipy.special import erfc, erf
from scipy.optimize import minimize
f32 = np.float32
x = np.linspace(0, 14, 280001)
NEGC = -np.log2(np.e)/2
e = np.exp2(NEGC*x*x)
Phim = 0.5*erfc(x/np.sqrt(2))
def errs(c):
    Q = np.polyval(np.r_[c[::-1], 0.5], x)
    d = np.abs(e*Q - Phim)
    return (x*d).max(), d.max()
def obj(c): return max(errs(c))
K = np.sqrt(-NEGC)
cz = np.array([-4.661208568e-01, 3.198611566e-01, -1.510052848e-01, 4.131927638e-02, -4.772022745e-03])
c0 = cz*K**np.arange(1,6)
r = minimize(obj, c0, method="Nelder-Mead", options=dict(xatol=1e-13, fatol=1e-15, maxiter=400000, maxfev=400000))
r = minimize(obj, r.x, method="Powell", options=dict(xtol=1e-13, ftol=1e-15))
r = minimize(obj, r.x, method="Nelder-Mead", options=dict(xatol=1e-13, fatol=1e-15, maxiter=400000, maxfev=400000))
print("coef:", ", ".join("%.9ef" % v for v in r.x), errs(r.x))
c = r.x.astype(f32)
# float32 emulation of the device sequence, both signs
xs = np.linspace(-14, 14, 560001).astype(f32)
def fma(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(f32)
ks = f32(1.0)
a = np.abs(xs)
arg = (xs*f32(NEGC))*xs
ee = np.exp2(arg.astype(np.float64)).astype(f32)
cc = [f32(0.5)] + [f32(v) for v in c]
q = fma(a, np.full_like(a, cc[5]), np.full_like(a, cc[4]))
for k in (3,2,1,0): q = fma(a, q, np.full_like(a, cc[k]))
us = fma(-ee, q, np.full_like(a, f32(0.5)))
cdf = f32(0.5) + np.copysign(us, xs)
y = fma(a, us, f32(0.5)*xs)
dy = fma(ee*xs, np.full_like(a, f32(0.3989422804)), cdf)
X = xs.astype(np.float64)
Phi = 0.5*(1+erf(X/np.sqrt(2))); phi = np.exp(-X*X/2)/np.sqrt(2*np.pi)
print("f32: gelu err %.3e  gelu' err %.3e  cdf err %.3e; min us %.3e" % (np.abs(y - X*Phi).max(), np.abs(dy - (Phi + X*phi)).max(), np.abs(cdf-Phi).max(), us.min()))
print("cdf range", cdf.min(), cdf.max(), " y at -14:", y[0], " at 14:", y[-1])
