"""Coefficients of the short atan used by the PLL wave: atan(z) = z + z*u*Q(u), u = z*z, |z| <= zmax.
Chebyshev interpolation of Q on [0, zmax^2] in 60-digit arithmetic, rounded to double; prints the C table
and the worst error of a plain double evaluation against mpmath (in ulps of the result)."""
import sys
import mpmath as mp
import numpy as np
mp.mp.dps = 60
zmax = mp.mpf(sys.argv[1]) if len(sys.argv) > 1 else mp.mpf("0.25")
ncoef = int(sys.argv[2]) if len(sys.argv) > 2 else 9
umax = zmax * zmax

def Q(u):
    if u == 0:
        return mp.mpf(-1) / 3
    z = mp.sqrt(u)
    return (mp.atan(z) / z - 1) / u

# Chebyshev nodes on [0, umax]
n = ncoef
nodes = [umax / 2 * (1 + mp.cos(mp.pi * (2 * k + 1) / (2 * n))) for k in range(n)]
A = mp.matrix(n, n)
b = mp.matrix(n, 1)
for i, x in enumerate(nodes):
    for j in range(n):
        A[i, j] = x ** j
    b[i] = Q(x)
c = mp.lu_solve(A, b)
cd = [float(ci) for ci in c]
print("static const double ATAN_Q[%d] = {" % n)
for v in cd:
    print("    %s," % float.hex(v) if False else "    %.17e," % v)
print("};")
# test
rng = np.random.default_rng(1)
zs = np.concatenate([rng.uniform(-float(zmax), float(zmax), 200000), np.linspace(-float(zmax), float(zmax), 20001), rng.uniform(-1e-3, 1e-3, 20000)])
u = zs * zs
q = np.zeros_like(zs)
for v in cd[::-1]:
    q = q * u + v
val = zs + zs * u * q
worst = 0
for z, v in zip(zs[::37], val[::37]):
    ex = mp.atan(mp.mpf(float(z)))
    if ex == 0:
        continue
    err = abs((mp.mpf(float(v)) - ex) / ex)
    worst = max(worst, float(err))
print("worst relative error %.3e (%.2f ulp of 2^-53)" % (worst, worst / 2 ** -53))
