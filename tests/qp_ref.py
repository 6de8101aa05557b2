"""Independent dense QP solver used to cross-check the oracle's Riccati IPM (tests only).

The stage QP exported by ``Oracle.build_qp`` is condensed onto the controls (states eliminated through the double
integrator), soft rows get an explicit slack variable, and the resulting inequality-only dense QP
    min 1/2 v^T H v + g^T v   s.t.  G v <= h
is solved by a plain long-step log-barrier/primal-dual Newton method with numpy.linalg -- no Riccati recursion, no
Mehrotra corrector, different variables: a different algorithm from the oracle's on purpose.
"""
import numpy as np


def condense(qp, N, nq, dt):
    nx, nu = 2 * nq, nq
    A = np.block([[np.eye(nq), dt * np.eye(nq)], [np.zeros((nq, nq)), np.eye(nq)]])
    Bm = np.vstack([0.5 * dt * dt * np.eye(nq), dt * np.eye(nq)])
    nv = N * nu
    # x_k = Phi_k v + c_k
    Phi = [np.zeros((nx, nv))]
    c = [qp['dx0'].copy()]
    for k in range(N):
        P = A @ Phi[k]
        P[:, k * nu:(k + 1) * nu] += Bm
        Phi.append(P)
        c.append(A @ c[k] + qp['b'][k][:nx])
    H = np.zeros((nv, nv)); g = np.zeros(nv)
    rowsG, rowsh, soft_w = [], [], []
    eqA, eqb = [], []
    for k in range(N + 1):
        nuk = nu if k < N else 0
        nz = nuk + nx
        # z_k = T v + t0
        T = np.zeros((nz, nv)); t0 = np.zeros(nz)
        if k < N:
            T[:nu, k * nu:(k + 1) * nu] = np.eye(nu)
        T[nuk:, :] = Phi[k]; t0[nuk:] = c[k]
        Hk = qp['H'][k][:nz, :nz]; gk = qp['g'][k][:nz]
        H += T.T @ Hk @ T
        g += T.T @ (Hk @ t0 + gk)
        for r in range(qp['nr'][k]):
            a = qp['C'][k][r, :nz]
            if qp['has_lo'][k][r] and qp['has_hi'][k][r] and qp['lo'][k][r] == qp['hi'][k][r]:
                eqA.append(a @ T); eqb.append(qp['lo'][k][r] - a @ t0)      # lb == ub: an equality row
                continue
            if qp['has_lo'][k][r]:
                rowsG.append(-(a @ T)); rowsh.append(-(qp['lo'][k][r] - a @ t0)); soft_w.append(qp['soft'][k][r])
            if qp['has_hi'][k][r]:
                rowsG.append(a @ T); rowsh.append(qp['hi'][k][r] - a @ t0); soft_w.append(-1.0)
    G = np.array(rowsG); h = np.array(rowsh); soft_w = np.array(soft_w)
    if eqA:
        # eliminate equalities through a null-space basis: v = vp + Z w
        from scipy.linalg import null_space
        Ae, be = np.array(eqA), np.array(eqb)
        vp = np.linalg.lstsq(Ae, be, rcond=None)[0]
        Z = null_space(Ae)
        return dict(H=Z.T @ H @ Z, g=Z.T @ (H @ vp + g), G=G @ Z, h=h - G @ vp, soft_w=soft_w, Z=Z, vp=vp,
                    Phi=Phi, c=c)
    return dict(H=H, g=g, G=G, h=h, soft_w=soft_w, Z=None, vp=None, Phi=Phi, c=c)


def solve_condensed(cq, **kw):
    w, s, lam, it = solve_dense(cq['H'], cq['g'], cq['G'], cq['h'], cq['soft_w'], **kw)
    v = w if cq['Z'] is None else cq['vp'] + cq['Z'] @ w
    return v, s, lam, it


def solve_dense(H, g, G, h, soft_w, tol=1e-11, max_iter=200):
    """rows with soft_w >= 0 are  G v - s <= h, s >= 0  with linear cost soft_w * s."""
    nv = H.shape[0]
    soft_idx = np.where(soft_w >= 0)[0]
    ns = len(soft_idx)
    n = nv + ns
    Hh = np.zeros((n, n)); Hh[:nv, :nv] = H
    gg = np.concatenate([g, soft_w[soft_idx]])
    m = G.shape[0]
    Gh = np.zeros((m + ns, n)); Gh[:m, :nv] = G
    hh = np.concatenate([h, np.zeros(ns)])
    for j, r in enumerate(soft_idx):
        Gh[r, nv + j] = -1.0
        Gh[m + j, nv + j] = -1.0           # -s <= 0
    v = np.zeros(n)
    v[nv:] = 1.0
    s = np.maximum(hh - Gh @ v, 1.0)
    lam = 1.0 / s
    for it in range(max_iter):
        rd = Hh @ v + gg + Gh.T @ lam
        rp = Gh @ v + s - hh
        mu = s @ lam / len(s)
        if max(np.abs(rd).max(), np.abs(rp).max(), mu) < tol:
            break
        sigma = 0.2
        rc = s * lam - sigma * mu
        D = lam / s
        K = Hh + Gh.T @ (D[:, None] * Gh)
        # derivation: ds = -rp - G dv ; dlam = (-rc - lam ds)/s
        dv = np.linalg.solve(K, -rd - Gh.T @ ((lam * rp - rc) / s))
        ds = -rp - Gh @ dv
        dlam = (-rc - lam * ds) / s
        a = 1.0
        for y, dy in ((s, ds), (lam, dlam)):
            neg = dy < 0
            if neg.any():
                a = min(a, 0.99 * np.min(-y[neg] / dy[neg]))
        v += a * dv; s += a * ds; lam += a * dlam
    return v[:nv], v[nv:], lam, it
