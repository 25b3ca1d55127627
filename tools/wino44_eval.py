#!/usr/bin/env python3
"""Numerics of Winograd F(4x4,3x3) against F(2x2,3x3) and the direct form, all in float32 against a float64
reference, on encoder-like data.  Pure numpy, CPU.  Prints normwise errors."""
import numpy as np
from fractions import Fraction as Fr


def cook_toom(points, m, r):
    """F(m, r) matrices AT [m x n], G [n x r], BT [n x n], n = m + r - 1, from n-1 finite points + infinity."""
    n = m + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == n - 1
    # Vandermonde-style construction (Lavin & Gray / wincnn): exact rationals
    def poly_mul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    # f_i = prod_{j != i} (p_i - p_j)
    f = []
    for i, pi in enumerate(pts):
        v = Fr(1)
        for j, pj in enumerate(pts):
            if i != j:
                v *= (pi - pj)
        f.append(v)
    AT = [[pts[j] ** i for j in range(n - 1)] + [Fr(1) if i == m - 1 else Fr(0)] for i in range(m)]
    G = [[pts[j] ** i / f[j] for i in range(r)] for j in range(n - 1)] + [[Fr(0)] * (r - 1) + [Fr(1)]]
    # BT rows: coefficients of prod_{j != i}(x - p_j) for finite points; last row: prod_j (x - p_j)
    BT = []
    for i in range(n - 1):
        poly = [Fr(1)]
        for j, pj in enumerate(pts):
            if j != i:
                poly = poly_mul(poly, [-pj, Fr(1)])
        BT.append(poly + [Fr(0)] * (n - len(poly)))
    poly = [Fr(1)]
    for pj in pts:
        poly = poly_mul(poly, [-pj, Fr(1)])
    BT.append(poly)
    tof = lambda M: np.array([[float(x) for x in row] for row in M], dtype=np.float64)
    return tof(AT), tof(G), tof(BT)


def check_1d(AT, G, BT, m, r):
    rs = np.random.RandomState(0)
    d, g = rs.randn(m + r - 1), rs.randn(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    return np.abs(y - ref).max()


def conv_direct(x, w, dt):
    # x [C, H, W], w [K, C, 3, 3] -> [K, H-2, W-2]; accumulate in dt over (c, ky, kx) like an fma chain
    C, H, W = x.shape
    K = w.shape[0]
    out = np.zeros((K, H - 2, W - 2), dtype=dt)
    for c in range(C):
        for ky in range(3):
            for kx in range(3):
                out += (w[:, c, ky, kx].astype(dt)[:, None, None] * x[c, ky:ky + H - 2, kx:kx + W - 2].astype(dt)[None]).astype(dt)
    return out


def conv_wino(x, w, AT, G, BT, m, dt):
    C, H, W = x.shape
    K = w.shape[0]
    n = m + 2
    Ho, Wo = H - 2, W - 2
    ty, tx = -(-Ho // m), -(-Wo // m)
    xp = np.zeros((C, ty * m + 2, tx * m + 2), dtype=dt)
    xp[:, :H, :W] = x
    ATd, Gd, BTd = AT.astype(dt), G.astype(dt), BT.astype(dt)
    U = np.einsum('ia,kcab,jb->kcij', Gd, w.astype(dt), Gd).astype(dt)          # [K,C,n,n]
    out = np.zeros((K, ty * m, tx * m), dtype=dt)
    for iy in range(ty):
        for ix in range(tx):
            d = xp[:, iy * m:iy * m + n, ix * m:ix * m + n]
            V = np.einsum('ia,cab,jb->cij', BTd, d, BTd).astype(dt)
            M = np.zeros((K, n, n), dtype=dt)
            for c in range(C):                                                  # fp32 accumulation over channels
                M += (U[:, c] * V[c][None]).astype(dt)
            Y = np.einsum('ia,kab,jb->kij', ATd, M, ATd).astype(dt)
            out[:, iy * m:iy * m + m, ix * m:ix * m + m] = Y
    return out[:, :Ho, :Wo]


def nerr(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))


def main():
    rs = np.random.RandomState(1)
    C = K = 32
    H = 21
    sets = {
        "F(2x2) pts 0,1,-1": ((0, 1, -1), 2),
        "F(4x4) pts 0,1,-1,2,-2": ((0, 1, -1, 2, -2), 4),
        "F(4x4) pts 0,1,-1,1/2,-1/2": ((0, 1, -1, Fr(1, 2), -Fr(1, 2)), 4),
        "F(4x4) pts 0,1,-1,1/2,-2": ((0, 1, -1, Fr(1, 2), -2), 4),
        "F(3x3) pts 0,1,-1,2": ((0, 1, -1, 2), 3),
        "F(3x3) pts 0,1,-1,1/2": ((0, 1, -1, Fr(1, 2)), 3),
    }
    for kind in ("relu activations (forward)", "signed gradients (dgrad-like)"):
        if kind.startswith("relu"):
            x = np.maximum(rs.randn(C, H, H), 0).astype(np.float32) * 0.7
        else:
            x = (rs.randn(C, H, H) * (rs.rand(C, H, H) < 0.5)).astype(np.float32)
        q, _ = np.linalg.qr(rs.randn(C * 9, K))
        w = (q.T * np.sqrt(2.0)).reshape(K, C, 3, 3).astype(np.float32)
        ref = conv_direct(x, w, np.float64)
        print(f"--- {kind}:  direct fp32 {nerr(conv_direct(x, w, np.float32), ref):.2e}")
        for name, (pts, m) in sets.items():
            AT, G, BT = cook_toom(pts, m, 3)
            assert check_1d(AT, G, BT, m, 3) < 1e-9, name
            e32 = nerr(conv_wino(x, w, AT, G, BT, m, np.float32), ref)
            e64 = nerr(conv_wino(x, w, AT, G, BT, m, np.float64), ref)
            y = conv_wino(x, w, AT, G, BT, m, np.float32).astype(np.float64)
            emax = float(np.abs(y - ref).max() / np.abs(ref).max())
            print(f"{name:34s} fp32 normwise {e32:.2e}  max/maxabs {emax:.2e}   (fp64 check {e64:.1e})")


if __name__ == "__main__":
    main()
