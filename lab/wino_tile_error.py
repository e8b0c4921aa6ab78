"""Numerical lead for a later round (CPU, numpy; nothing in the product uses it): how much accuracy would larger Winograd tiles cost?
A 6x6 map is four F(3x3,3x3) tiles today (100 transform-domain values and multiplies per channel pair).  F(6,3) along one axis and
F(3,3) along the other needs two 8x5 tiles = 80 (-20 % multiplies AND -20 % transform-domain traffic); F(6x6,3x3) one 8x8 tile = 64.
Emulated in float32 (transforms in float32, or in float64 rounded once; float32 products with sequential accumulation), one layer,
L-inf / L-inf against the float64 direct convolution.  Measured here (round 4): F(3x3) 2.3e-6, F(6x3) 4.1-5.3e-6, F(6x6) 1.0e-5 at 64
channels; 5.5e-6 / 1.2e-5 / 1.9e-5 at 384 channels with a pessimistic non-fused accumulation (direct: 2.7e-6) - i.e. F(6x3) costs
about 2.1x the error of today's tiles, inside the 1e-5 logit tolerance with a margin of ~2, F(6x6) does not fit.  Points 0, +-1, +-2,
+-1/2, inf; a point set with +-1/4 is 10-100x worse.
"""
import numpy as np, itertools
from fractions import Fraction as F

def cook_toom(m, r, pts):
    """1-D Winograd F(m, r) matrices (AT [m x n], G [n x r], BT [n x n]) with n = m + r - 1 points incl. infinity (last), exact rationals."""
    n = m + r - 1
    assert len(pts) == n - 1
    pts = [F(p) for p in pts]
    # polynomial interpolation construction (Lavin): AT = Vandermonde^T (m rows), G = Vandermonde (r cols) scaled, BT from inverse
    # Use the standard: Y = AT [(G g) * (BT d)]
    # Build via transposed Toom-Cook: let V_k be the n x k matrix with rows [1, p, p^2, ...] for finite points and [0..0,1] for infinity
    def V(k):
        M = [[p ** j for j in range(k)] for p in pts]
        M.append([F(0)] * (k - 1) + [F(1)])
        return M
    Vn = V(n)       # n x n
    # inverse of Vn^T? Standard derivation: linear convolution of length m (kernel... ) We use the FIR form: y = AT[(G g) . (BT d)],
    # AT = V(m)^T, G = diag(1/N_i) V(r), BT = (Vn^-T) scaled ... derive: BT = inverse(Vn)^T * diag(N)?  Solve numerically in rationals:
    import sympy as sp
    Vn_s = sp.Matrix(Vn)
    inv = Vn_s.inv()           # n x n
    AT = sp.Matrix(V(m)).T     # m x n
    G = sp.Matrix(V(r))        # n x r
    BT = inv.T                 # n x n   (then y = AT [ (G g) . (BT d) ] holds for the transposed Toom-Cook)
    return np.array(AT.tolist(), dtype=np.float64), np.array(G.tolist(), dtype=np.float64), np.array(BT.tolist(), dtype=np.float64)

def check(m, r, pts):
    AT, G, BT = cook_toom(m, r, pts)
    rng = np.random.default_rng(0)
    d = rng.standard_normal(m + r - 1); g = rng.standard_normal(r)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([sum(d[i + k] * g[k] for k in range(r)) for i in range(m)])
    return np.abs(y - ref).max()

def conv_err(mr, mc, pts_r, pts_c, C=64, K=64, N=8, seed=1, scale_in=1.0):
    """6x6 maps, 3x3 conv pad 1; tiles of mr x mc outputs; transforms + products in float32; returns relmax error vs float64 direct."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((N, C, 6, 6)) * scale_in).astype(np.float32)
    w = (rng.standard_normal((K, C, 3, 3)) * np.sqrt(2.0 / (C * 9 + K * 9))).astype(np.float32)
    xp = np.zeros((N, C, 8, 8)); xp[:, :, 1:7, 1:7] = x
    ref = np.zeros((N, K, 6, 6))
    for a in range(3):
        for b in range(3):
            ref += np.einsum('nchw,kc->nkhw', xp[:, :, a:a + 6, b:b + 6], w[:, :, a, b].astype(np.float64))
    ATr, Gr, BTr = cook_toom(mr, 3, pts_r); ATc, Gc, BTc = cook_toom(mc, 3, pts_c)
    f = np.float32
    U = np.einsum('ia,kcab,jb->kcij', Gr.astype(f), w, Gc.astype(f)).astype(f)          # [K,C,nr,nc]
    out = np.zeros((N, K, 6, 6), dtype=f)
    nr, nc = mr + 2, mc + 2
    for ty in range(6 // mr):
        for tx in range(6 // mc):
            dwin = xp[:, :, ty * mr:ty * mr + nr, tx * mc:tx * mc + nc].astype(f)
            V = np.einsum('ia,ncab,jb->ncij', BTr.astype(f), dwin, BTc.astype(f)).astype(f)
            M = np.einsum('ncij,kcij->nkij', V, U).astype(f)      # float32 accumulate (numpy pairwise-ish)
            Y = np.einsum('ia,nkab,jb->nkij', ATr.astype(f), M, ATc.astype(f)).astype(f)
            out[:, :, ty * mr:(ty + 1) * mr, tx * mc:(tx + 1) * mc] = Y
    return np.abs(out - ref).max() / np.abs(ref).max()

if __name__ == "__main__":
    p33 = [0, 1, -1, 2]
    print("exactness F(3,3):", check(3, 3, p33))
    for name, p63 in (("0,1,-1,2,-2,1/2,-1/2", [0, 1, -1, 2, -2, F(1, 2), F(-1, 2)]),
                      ("0,1,-1,1/2,-1/2,2,-2 (same)", [0, 1, -1, F(1, 2), F(-1, 2), 2, -2]),
                      ("0,1,-1,1/2,-1/2,3/2,-3/2", [0, 1, -1, F(1, 2), F(-1, 2), F(3, 2), F(-3, 2)]),
                      ("0,1,-1,2/3,-2/3,3/2,-3/2", [0, 1, -1, F(2, 3), F(-2, 3), F(3, 2), F(-3, 2)]),
                      ("0,1,-1,1/2,-1/2,1/4,-1/4", [0, 1, -1, F(1, 2), F(-1, 2), F(1, 4), F(-1, 4)])):
        print("exactness F(6,3)", name, check(6, 3, p63))
        e33 = np.mean([conv_err(3, 3, p33, p33, seed=s) for s in range(3)])
        e63 = np.mean([conv_err(6, 3, p63, p33, seed=s) for s in range(3)])
        e66 = np.mean([conv_err(6, 6, p63, p63, seed=s) for s in range(3)])
        print(f"  relmax err: F(3x3) {e33:.2e}   F(6x3) {e63:.2e}   F(6x6) {e66:.2e}")

def conv_err64(mr, mc, pts_r, pts_c, C=64, K=64, N=8, seed=1, Kacc="f32"):
    """transforms evaluated in float64 and rounded ONCE to float32 (U, V); products / accumulation float32; output transform float64"""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, C, 6, 6)).astype(np.float32)
    w = (rng.standard_normal((K, C, 3, 3)) * np.sqrt(2.0 / (C * 9 + K * 9))).astype(np.float32)
    xp = np.zeros((N, C, 8, 8)); xp[:, :, 1:7, 1:7] = x
    ref = np.zeros((N, K, 6, 6))
    for a in range(3):
        for b in range(3):
            ref += np.einsum('nchw,kc->nkhw', xp[:, :, a:a + 6, b:b + 6], w[:, :, a, b].astype(np.float64))
    ATr, Gr, BTr = cook_toom(mr, 3, pts_r); ATc, Gc, BTc = cook_toom(mc, 3, pts_c)
    f = np.float32
    U = np.einsum('ia,kcab,jb->kcij', Gr, w.astype(np.float64), Gc).astype(f)
    out = np.zeros((N, K, 6, 6))
    nr, nc = mr + 2, mc + 2
    for ty in range(6 // mr):
        for tx in range(6 // mc):
            dwin = xp[:, :, ty * mr:ty * mr + nr, tx * mc:tx * mc + nc]
            V = np.einsum('ia,ncab,jb->ncij', BTr, dwin, BTc).astype(f)
            M = np.zeros((N, K, nr, nc), dtype=f)
            for c in range(C):                                   # sequential float32 fma-like accumulation over channels
                M = (M + V[:, None, c] * U[None, :, c]).astype(f)
            Y = np.einsum('ia,nkab,jb->nkij', ATr, M.astype(np.float64), ATc)
            out[:, :, ty * mr:(ty + 1) * mr, tx * mc:(tx + 1) * mc] = Y
    return np.abs(out.astype(f) - ref).max() / np.abs(ref).max()

if __name__ == "__main__":
    print("--- float64 transforms, float32 products, sequential float32 accumulation over 64 channels")
    p33 = [0, 1, -1, 2]
    p63 = [0, 1, -1, 2, -2, F(1, 2), F(-1, 2)]
    for C in (64, 384):
        e33 = np.mean([conv_err64(3, 3, p33, p33, C=C, K=32, N=4, seed=s) for s in range(2)])
        e63 = np.mean([conv_err64(6, 3, p63, p33, C=C, K=32, N=4, seed=s) for s in range(2)])
        e66 = np.mean([conv_err64(6, 6, p63, p63, C=C, K=32, N=4, seed=s) for s in range(2)])
        print(f"C={C}: F(3x3) {e33:.2e}   F(6x3) {e63:.2e}   F(6x6) {e66:.2e}")
    # direct conv in float32 with sequential accumulation for comparison
    for C in (64, 384):
        rng = np.random.default_rng(1)
        x = rng.standard_normal((4, C, 6, 6)).astype(np.float32)
        w = (rng.standard_normal((32, C, 3, 3)) * np.sqrt(2.0 / (C * 9 + 32 * 9))).astype(np.float32)
        xp = np.zeros((4, C, 8, 8), dtype=np.float32); xp[:, :, 1:7, 1:7] = x
        ref = np.zeros((4, 32, 6, 6)); acc = np.zeros((4, 32, 6, 6), dtype=np.float32)
        for c in range(C):
            for a in range(3):
                for b in range(3):
                    t = xp[:, c, a:a + 6, b:b + 6][:, None] * w[None, :, c, a, b][:, :, None, None]
                    acc = (acc + t).astype(np.float32); ref += t.astype(np.float64)
        print(f"C={C}: direct float32 sequential {np.abs(acc - ref).max() / np.abs(ref).max():.2e}")
