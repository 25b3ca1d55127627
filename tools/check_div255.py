"""Exhaustive check (CPU, numpy) that  q = v*r; e = fma(-q, 255, v); q' = fma(e, r, q)  with r = RN(1/255) equals the
correctly rounded float32 quotient v / 255 for EVERY float32 mantissa (one binade; the identity is invariant under
scaling v by powers of two while nothing leaves the normal range -- the augmentation's values lie in [0, 256)).
The aug kernel uses this 3-instruction sequence instead of the ~10-instruction IEEE division."""
import numpy as np
from fractions import Fraction

r32 = np.float32(1.0) / np.float32(255.0)
m = np.arange(1 << 23, 1 << 24, dtype=np.int64)
v = (m.astype(np.float64) * 2.0 ** -23)                     # exact: every float32 in [1, 2)
# reference: correctly rounded quotient by integer arithmetic (255 is odd: no ties)
k = np.where(m * 256 // 255 < (1 << 24), 8, 7).astype(np.int64)
num = m << k
Q = num // 255
rem = num - Q * 255
Q = Q + (2 * rem > 255)
ref = Q.astype(np.float64) * np.exp2(-(23 + k).astype(np.float64))      # exact
assert np.all(ref.astype(np.float32).astype(np.float64) == ref)
# candidate
v32 = v.astype(np.float32)
q0 = (v32 * r32).astype(np.float32)                          # RN32(v * r)
e = v - q0.astype(np.float64) * 255.0                        # exact in double (see header comment in the source)
assert np.all(e.astype(np.float32).astype(np.float64) == e)  # ... and representable: the fma rounds nothing
p = e * float(r32)                                           # exact in double (<= 34 significant bits)
ulp = np.spacing(q0).astype(np.float64)                      # ulp of q0 (upwards)
t = p / ulp
q1 = q0.astype(np.float64) + np.where(t > 0.5, ulp, 0.0) - np.where(t < -0.5, ulp, 0.0)
hard = (np.abs(np.abs(t) - 0.5) < 1e-6) | (np.frexp(q0.astype(np.float64))[0] == 0.5)
print("elements needing the exact fallback:", int(hard.sum()))
for i in np.nonzero(hard)[0]:
    s = Fraction(float(q0[i])) + Fraction(float(e[i])) * Fraction(float(r32))
    lo = np.float32(float(s))                                # float(Fraction) is correctly rounded to double; then
    cands = [np.float32(lo), np.nextafter(np.float32(lo), np.float32(0)), np.nextafter(np.float32(lo), np.float32(1))]
    best = min(cands, key=lambda c: (abs(Fraction(float(c)) - s), int(np.float32(c).view(np.uint32)) & 1))
    q1[i] = float(best)
bad = np.nonzero(q1 != ref)[0]
print("mantissas checked:", m.size, " mismatches:", bad.size)
if bad.size:
    print("first:", v[bad[:5]], q1[bad[:5]], ref[bad[:5]])
    raise SystemExit(1)
