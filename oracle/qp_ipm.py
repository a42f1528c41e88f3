"""Dense primal-dual interior-point QP solver + KKT certificate -- ORACLE (test infrastructure).

Stands in for the conic interior-point solver (ECOS / Clarabel, un-pinned) that
``cvxpy.Problem.solve`` drops into at reference ``tzddpc/tzddpc.py:367``.  Like those solvers it
returns the analytic-centre optimum when the optimum is not unique.

    minimise 1/2 x'Px + q'x   subject to   l <= A x <= u      (entries of l/u may be -/+inf)

No solver is trusted: every solution used as a golden is accepted only through
``kkt_certificate`` (primal / dual residual and complementarity), which is independent of how
the point was obtained.
"""
from __future__ import annotations

import numpy as np


class QPResult(dict):
    __getattr__ = dict.get


def kkt_certificate(P, q, A, l, u, x, y):
    """Scaled KKT residuals of (x, y) for  min 1/2x'Px+q'x, l<=Ax<=u  (y>0 upper, y<0 lower)."""
    Ax = A @ x
    Px = P @ x
    Aty = A.T @ y
    old = np.seterr(invalid="ignore")
    prim = max(0.0, float(np.max(np.maximum(l - Ax, 0.0), initial=0.0)), float(np.max(np.maximum(Ax - u, 0.0), initial=0.0)))
    dual = float(np.max(np.abs(Px + q + Aty), initial=0.0))
    yp, ym = np.maximum(y, 0.0), np.minimum(y, 0.0)
    gap_u = np.where(np.isfinite(u), yp * (u - Ax), np.where(yp > 0, np.inf, 0.0))
    gap_l = np.where(np.isfinite(l), ym * (Ax - l), np.where(ym < 0, np.inf, 0.0))
    comp = float(max(np.max(np.abs(gap_u), initial=0.0), np.max(np.abs(gap_l), initial=0.0)))
    sp = 1.0 + max(float(np.max(np.abs(Ax), initial=0.0)), float(np.max(np.abs(x), initial=0.0)))
    sd = 1.0 + max(float(np.max(np.abs(Px), initial=0.0)), float(np.max(np.abs(Aty), initial=0.0)), float(np.max(np.abs(q), initial=0.0)))
    obj = 0.5 * x @ Px + q @ x
    np.seterr(**old)
    return dict(primal=float(prim / sp), dual=float(dual / sd), comp=float(comp / (1.0 + abs(obj))), obj=float(obj))


def solve_qp(P, q, A, l, u, tol: float = 1e-10, max_iter: int = 100, verbose: bool = False) -> QPResult:
    P = np.asarray(P, float); q = np.asarray(q, float); A = np.asarray(A, float)
    l = np.asarray(l, float).copy(); u = np.asarray(u, float).copy()
    n = q.size
    A_full, l_full, u_full = A, l.copy(), u.copy()
    # rows without any coefficient are pure-parameter tests (e.g. xbar0 + e0 in X at k = 0)
    zero_row = np.abs(A).max(axis=1, initial=0.0) == 0.0
    if np.any(zero_row):
        if np.any(l[zero_row] > 1e-12) or np.any(u[zero_row] < -1e-12):
            return QPResult(x=np.full(n, np.nan), y=np.zeros(A.shape[0]), status="infeasible", iters=0, obj=np.inf,
                            cert=dict(primal=np.inf, dual=np.inf, comp=np.inf, obj=np.inf))
        A = A[~zero_row]; l = l[~zero_row]; u = u[~zero_row]
    eq = np.isfinite(l) & np.isfinite(u) & (np.abs(u - l) <= 1e-13 * (1 + np.abs(u)))
    up = np.isfinite(u) & ~eq
    lo = np.isfinite(l) & ~eq
    E, f = A[eq], u[eq]
    G = np.vstack([A[up], -A[lo]])
    h = np.concatenate([u[up], -l[lo]])
    mi, me = h.size, f.size

    # row scaling of the inequality block for conditioning
    gs = np.maximum(np.linalg.norm(G, axis=1), 1e-12) if mi else np.ones(0)
    G = G / gs[:, None]; h = h / gs

    # cost normalisation (losses with 1e9 weights, reference examples/3.5dimsystem_sim.py:19)
    cscale = 1.0 / max(1.0, float(np.abs(q).max(initial=0.0)), float(np.abs(P).max(initial=0.0)))
    P_in, q_in = P, q
    P = P * cscale; q = q * cscale

    x = np.zeros(n); y = np.zeros(me)
    s = np.ones(mi); z = np.ones(mi)

    def solve_kkt(w, r1, r2):
        """[P + G'diag(w)G + eps, E'; E, -eps] [dx; dy] = [r1; r2] with refinement."""
        H = P + (G.T * w) @ G if mi else P.copy()
        reg = 1e-13 * (1.0 + np.abs(np.diag(P)).max(initial=0.0))
        K = np.zeros((n + me, n + me))
        K[:n, :n] = H + reg * np.eye(n)
        if me:
            K[:n, n:] = E.T; K[n:, :n] = E
            K[n:, n:] = -reg * np.eye(me)
        rhs = np.concatenate([r1, r2])
        try:
            import scipy.linalg as sla
            lu = sla.lu_factor(K)
            sol = sla.lu_solve(lu, rhs)
            K0 = K.copy(); K0[:n, :n] -= reg * np.eye(n)
            if me:
                K0[n:, n:] = 0.0
            for _ in range(5):
                sol = sol + sla.lu_solve(lu, rhs - K0 @ sol)
        except Exception:
            sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
        return sol[:n], sol[n:]

    if mi:
        # CVXOPT-style start: least-squares point, then shift s, z into the cone
        x, y = solve_kkt(np.ones(mi), -q + G.T @ h, f)
        r = h - G @ x
        s = r + max(0.0, 1.0 - r.min()) if r.min() <= 1e-8 else r
        z = np.ones(mi)
    status = "max_iter"
    np_err = np.seterr(divide="ignore", invalid="ignore", over="ignore")
    for it in range(max_iter):
        rd = P @ x + q + (G.T @ z if mi else 0.0) + (E.T @ y if me else 0.0)
        rp = G @ x + s - h if mi else np.zeros(0)
        re = E @ x - f if me else np.zeros(0)
        mu = float(s @ z) / mi if mi else 0.0
        sc_d = 1.0 + max(np.abs(q).max(initial=0.0), np.abs(P @ x).max(initial=0.0))
        sc_p = 1.0 + max(np.abs(h).max(initial=0.0), np.abs(f).max(initial=0.0))
        if verbose:
            print(f"it {it:3d} rd {np.abs(rd).max():.2e} rp {np.abs(rp).max(initial=0):.2e} re {np.abs(re).max(initial=0):.2e} mu {mu:.2e}")
        if (np.abs(rd).max(initial=0.0) <= tol * sc_d and np.abs(rp).max(initial=0.0) <= tol * sc_p
                and np.abs(re).max(initial=0.0) <= tol * sc_p and mu <= tol):
            status = "solved"
            break
        if mi and mu <= 1e-3 * tol:
            ok = (np.abs(rd).max(initial=0.0) <= 1e3 * tol * sc_d and np.abs(rp).max(initial=0.0) <= 1e3 * tol * sc_p
                  and np.abs(re).max(initial=0.0) <= 1e3 * tol * sc_p)
            status = "solved" if ok else "stalled"
            break
        if mi and (np.abs(x).max(initial=0.0) > 1e13 or z.max() > 1e16):
            status = "infeasible_or_unbounded"
            break
        w = z / s if mi else np.ones(0)

        def newton(rc):
            # rc: target for s*dz + z*ds = -rc
            r1 = -rd - (G.T @ ((-rc + z * rp) / s) if mi else 0.0)
            dx, dy = solve_kkt(w, r1, -re)
            if mi:
                ds = -rp - G @ dx
                dz = (-rc - z * ds) / s
            else:
                ds = dz = np.zeros(0)
            return dx, dy, ds, dz

        def max_step(v, dv):
            neg = dv < 0
            return min(1.0, float(np.min(-v[neg] / dv[neg]))) if np.any(neg) else 1.0

        dx_a, dy_a, ds_a, dz_a = newton(s * z)
        if mi:
            a_p = max_step(s, ds_a); a_d = max_step(z, dz_a)
            mu_aff = float((s + a_p * ds_a) @ (z + a_d * dz_a)) / mi
            sigma = (mu_aff / mu) ** 3 if mu > 0 else 0.0
            dx, dy, ds, dz = newton(s * z + ds_a * dz_a - sigma * mu)
            a_p = min(1.0, 0.995 * max_step(s, ds)) if max_step(s, ds) < 1.0 else 1.0
            a_d = min(1.0, 0.995 * max_step(z, dz)) if max_step(z, dz) < 1.0 else 1.0
            a = min(a_p, a_d)
            x = x + a * dx; y = y + a * dy; s = s + a * ds; z = z + a * dz
        else:
            x = x + dx_a; y = y + dy_a

    np.seterr(**np_err)
    # map multipliers back to the l <= Ax <= u convention
    yy_full = np.zeros(A_full.shape[0])
    yy = np.zeros(A.shape[0])
    zu = z[: int(up.sum())] / gs[: int(up.sum())] if mi else np.zeros(0)
    zl = z[int(up.sum()):] / gs[int(up.sum()):] if mi else np.zeros(0)
    yy[up] += zu / cscale
    yy[lo] -= zl / cscale
    yy[eq] = y / cscale
    P, q = P_in, q_in
    yy_full[~zero_row] = yy
    A, l, u, yy = A_full, l_full, u_full, yy_full
    with np.errstate(invalid="ignore", divide="ignore"):
        cert = kkt_certificate(P, q, A, l, u, x, yy)
    return QPResult(x=x, y=yy, status=status, iters=it, obj=cert["obj"], cert=cert)
