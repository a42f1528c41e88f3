"""numpy restatement of the per-step numeric path of the HIP kernels -- ORACLE, test infrastructure.

The algorithm of ``tzddpc_amd/csrc`` (``tz_prepare`` -> ``tz_ipm``) written with plain numpy so
kernel results can be compared quantity by quantity.  It stands where the reference calls
``self.problem_full.solve(**cvxpy_kwargs)`` (reference ``tzddpc/tzddpc.py:367``), i.e. a conic
interior-point solver.  The independent check of the *answer* is ``oracle.qp_ipm`` (different
formulation, LU with pivoting, own start point) plus KKT certificates.

Device form: all rows one-sided and scaled,

    min 1/2 x'Px + q'x   s.t.  G x + s = h,  s >= 0          (lambda >= 0 multipliers)

Mehrotra predictor-corrector; the Newton system is reduced to  (P + G' diag(lambda/s) G + reg I) dx = r
and solved by Cholesky WITHOUT pivoting (what the MFMA kernel does), with one step of iterative
refinement.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class IpmOptions:
    max_iter: int = 40
    tol: float = 1e-9          # scaled residual / gap tolerance
    reg: float = 1e-13         # static primal regularisation (absolute; the scaled problem has O(1) entries)
    step_frac: float = 0.99999
    scaling_iters: int = 15
    refine: int = 2


def ruiz(P, G, q_ref, iters):
    nz, mi = P.shape[0], G.shape[0]
    D = np.ones(nz); E = np.ones(mi)
    Ps, Gs = P.copy(), G.copy()
    for _ in range(iters):
        cn = np.maximum(np.abs(Ps).max(axis=0, initial=0.0), np.abs(Gs).max(axis=0, initial=0.0))
        rn = np.abs(Gs).max(axis=1, initial=0.0)
        cn = np.where(cn < 1e-8, 1.0, cn); rn = np.where(rn < 1e-8, 1.0, rn)
        d = 1.0 / np.sqrt(cn); e = 1.0 / np.sqrt(rn)
        Ps = d[:, None] * Ps * d[None, :]; Gs = e[:, None] * Gs * d[None, :]
        D *= d; E *= e
    qn = np.abs(D * q_ref).max(initial=0.0)
    pn = np.abs(Ps).max(initial=0.0)
    c = 1.0 / max(pn, qn, 1e-300)
    return D, E, c


class Setup:
    """One-sided scaled form of a ``tzddpc_amd.builder.ParametricQP`` (host, build time)."""

    def __init__(self, qp, opt: IpmOptions = IpmOptions()):
        self.qp, self.opt = qp, opt
        fu = np.isfinite(qp.u0); fl = np.isfinite(qp.l0)
        self.G_un = np.vstack([qp.A[fu], -qp.A[fl]])
        self.h0 = np.concatenate([qp.u0[fu], -qp.l0[fl]])
        self.Ht = np.vstack([qp.Ut[fu], -qp.Lt[fl]])
        self.row_of = np.concatenate([np.nonzero(fu)[0], np.nonzero(fl)[0]])
        self.sign = np.concatenate([np.ones(int(fu.sum())), -np.ones(int(fl.sum()))])
        q_ref = np.abs(qp.q0) + np.abs(qp.Qt).sum(axis=1)
        self.D, self.E, self.c = ruiz(qp.P, self.G_un, q_ref, opt.scaling_iters)
        self.P = self.c * self.D[:, None] * qp.P * self.D[None, :]
        self.G = self.E[:, None] * self.G_un * self.D[None, :]

    def vectors(self, theta):
        qp = self.qp
        q = self.c * self.D * (qp.q0 + qp.Qt @ theta)
        h = self.E * (self.h0 + self.Ht @ theta)
        pv = qp.f0 + qp.Ft @ theta
        feas = bool(np.all(pv >= qp.pl - 1e-9) and np.all(pv <= qp.pu + 1e-9))
        return q, h, feas


def chol_nopivot(H):
    n = H.shape[0]
    L = np.tril(H).copy()
    for j in range(n):
        d = L[j, j]
        if not d > 0:
            return None
        d = np.sqrt(d)
        L[j, j] = d
        L[j + 1:, j] /= d
        L[j + 1:, j + 1:] -= np.tril(np.outer(L[j + 1:, j], L[j + 1:, j]))
    return L


def ipm(S: Setup, q, h, log=None):
    o = S.opt
    P, G = S.P, S.G
    nz, mi = P.shape[0], G.shape[0]
    import scipy.linalg as sla

    def factor(w):
        H = P + (G.T * w) @ G
        H[np.diag_indices(nz)] += o.reg
        L = np.linalg.cholesky(H)
        return H, L

    def solve(H, L, r):
        x = sla.cho_solve((L, True), r)
        for _ in range(o.refine):
            x = x + sla.cho_solve((L, True), r - H @ x)
        return x

    # start (CVXOPT style): least-squares point with w = 1, then shift into the cone
    H, L = factor(np.ones(mi))
    x = solve(H, L, -q + G.T @ h)
    r = h - G @ x
    s = r + max(0.0, 1.0 - r.min()) if r.min() <= 1e-8 else r
    lam = np.ones(mi)
    sc_d = 1.0 + np.abs(q).max(initial=0.0)
    sc_p = 1.0 + np.abs(h).max(initial=0.0)
    status = 1   # max_iter
    it = 0
    for it in range(o.max_iter):
        rd = P @ x + q + G.T @ lam
        rp = G @ x + s - h
        mu = float(s @ lam) / mi
        nrd = np.abs(rd).max(initial=0.0) / sc_d; nrp = np.abs(rp).max(initial=0.0) / sc_p
        if log is not None:
            log.append((it, nrd, nrp, mu))
        if nrd <= o.tol and nrp <= o.tol and mu <= o.tol:
            status = 0
            break
        if mu <= 1e-3 * o.tol:          # complementarity exhausted: accept if residuals are near tolerance
            status = 0 if (nrd <= 1e3 * o.tol and nrp <= 1e3 * o.tol) else 3
            break
        if not np.isfinite(mu) or np.abs(x).max(initial=0.0) > 1e14:
            status = 2
            break
        w = lam / s
        H, L = factor(w)

        def newton(rc):
            r1 = -rd - G.T @ ((-rc + lam * rp) / s)
            dx = solve(H, L, r1)
            ds = -rp - G @ dx
            dl = (-rc - lam * ds) / s
            return dx, ds, dl

        def max_step(v, dv):
            neg = dv < 0
            return min(1.0, float(np.min(-v[neg] / dv[neg]))) if np.any(neg) else 1.0

        dxa, dsa, dla = newton(s * lam)
        ap = max_step(s, dsa); ad = max_step(lam, dla)
        mu_aff = float((s + ap * dsa) @ (lam + ad * dla)) / mi
        sigma = (mu_aff / mu) ** 3
        dx, ds, dl = newton(s * lam + dsa * dla - sigma * mu)
        a = min(1.0, o.step_frac * min(max_step(s, ds), max_step(lam, dl)))
        x = x + a * dx; s = s + a * ds; lam = lam + a * dl
    return x, s, lam, status, it


def unscale(S: Setup, x, lam):
    """-> z (unscaled decision vector), y per original row (l <= Az <= u convention)."""
    z = S.D * x
    y_rows = S.E * lam / S.c
    y = np.zeros(S.qp.nc)
    np.add.at(y, S.row_of, S.sign * y_rows)
    return z, y
