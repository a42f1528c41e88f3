"""TEST INFRASTRUCTURE -- CPU restatement of ``TZDDPC.solve_simplified2`` (reference ``tzddpc/tzddpc.py:381-500``).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product never does.

The reference's problem is written out variable for variable and constraint for constraint (nothing condensed, nothing
eliminated) and handed to the oracle's interior point (``oracle/qp_ipm.py``, equality rows native):

    xi = [ xbar (N+1) x n | v N x m | ubar N x m | beta_x N x gamma_X | beta_u N x gamma_U ]        (+ loss epigraphs)

Parity unpinned, two [unverified-dep] points (pyzonotope is not vendored):
  * ``Ze[-1].sum()`` (``:451``) -- not defined anywhere in the reference repository.  ``ze_sum="radius"`` (default) takes it as
    the row sums of the absolute generators (the quantity ``.interval`` is built from: a size penalty, which is what the variable
    name ``regularizer`` says); ``ze_sum="columns"`` as the row sums of the whole matrix ``Z = [c | G]``.
  * ``term_2 @ Acl`` (``:459``) with ``term_2`` a length-n vector expression is the vector-matrix product ``Acl' term_2``.
"""
from __future__ import annotations

import numpy as np

from .qp_ipm import solve_qp


def build(A, B, K, deltaA, deltaB, W, X, U, Zsigma, N, xbar0, e0, loss, constraints=None, ze_sum="radius"):
    """Returns dict(P, q, r, A, l, u, ...) over xi as above; W, X, U, Zsigma[k] oracle.zonolite.Zonotope."""
    A = np.asarray(A, float); B = np.asarray(B, float); K = np.atleast_2d(np.asarray(K, float))
    n, m = B.shape
    assert len(Zsigma) == N, "Zsigma needs to be a list of zonotopes of length == N, the horizon"       # :405
    gx, gu = X.generators.shape[1], U.generators.shape[1]
    o_xb, o_v = 0, (N + 1) * n
    o_ub = o_v + N * m
    o_bx = o_ub + N * m
    o_bu = o_bx + N * gx
    nxi = o_bu + N * gu
    ixb = lambda k, i: o_xb + k * n + i
    ixv = lambda k, j: o_v + k * m + j
    iub = lambda k, j: o_ub + k * m + j
    Acl = A + B @ K                                                                  # :414
    rows, lo, hi = [], [], []

    def add(r, l_, h_):
        rows.append(r); lo.append(l_); hi.append(h_)

    # :419-428
    for k in range(N):
        for g in range(gu):
            r = np.zeros(nxi); r[o_bu + k * gu + g] = 1.0; add(r, -1.0, 1.0)
        for g in range(gx):
            r = np.zeros(nxi); r[o_bx + k * gx + g] = 1.0; add(r, -1.0, 1.0)
    for k in range(N):
        for j in range(m):                                                           # ubar == U.center + beta_u U.G'
            r = np.zeros(nxi); r[iub(k, j)] = 1.0; r[o_bu + k * gu:o_bu + (k + 1) * gu] = -U.generators[j]
            add(r, U.center[j], U.center[j])
        for i in range(n):                                                           # xbar[1:] == X.center + beta_x X.G'
            r = np.zeros(nxi); r[ixb(k + 1, i)] = 1.0; r[o_bx + k * gx:o_bx + (k + 1) * gx] = -X.generators[i]
            add(r, X.center[i], X.center[i])
    for i in range(n):                                                               # xbar[0] == xbar0
        r = np.zeros(nxi); r[ixb(0, i)] = 1.0; add(r, xbar0[i], xbar0[i])
    for k in range(N):                                                               # ubar == xbar[:-1] K' + v
        for j in range(m):
            r = np.zeros(nxi); r[iub(k, j)] = 1.0; r[ixv(k, j)] = -1.0
            r[[ixb(k, c) for c in range(n)]] -= K[j]
            add(r, 0.0, 0.0)
    # tube: Ze[k] = <centre_const + Ccoef xi, G>
    cen0, cenC, G = np.asarray(e0, float).copy(), np.zeros((n, nxi)), np.zeros((n, 1))        # :430
    t1 = W + Zsigma[0]                                                               # :432
    term1 = [t1]
    for k in range(1, N):                                                            # :435-436   Z * M = M @ Z
        term1.append(term1[-1] * Acl + (W + Zsigma[k]))
    t2_0, t2_C = np.zeros(n), np.zeros((n, nxi))                                     # :433
    Xi, Ui = X.interval, U.interval
    reg0, regC = 0.0, np.zeros(nxi)
    ze = [(cen0.copy(), cenC.copy(), G.copy())]
    for k in range(N):
        radx = np.abs(G).sum(axis=1); radu = np.abs(K @ G).sum(axis=1)
        for i in range(n):                                                           # :446-447
            r = cenC[i].copy(); r[ixb(k, i)] += 1.0
            add(r, Xi.left_limit[i] - cen0[i] + radx[i], Xi.right_limit[i] - cen0[i] - radx[i])
        Kc0, KcC = K @ cen0, K @ cenC
        for j in range(m):                                                           # :448-449
            r = KcC[j].copy(); r[iub(k, j)] += 1.0
            add(r, Ui.left_limit[j] - Kc0[j] + radu[j], Ui.right_limit[j] - Kc0[j] - radu[j])
        for i in range(n):                                                           # :445  xbar[k+1] == Acl xbar[k] + B v[k]
            r = np.zeros(nxi); r[ixb(k + 1, i)] = 1.0
            r[[ixb(k, c) for c in range(n)]] -= Acl[i]
            r[[ixv(k, c) for c in range(m)]] -= B[i]
            add(r, 0.0, 0.0)
        if ze_sum == "radius":                                                       # :451
            reg0 += float(radx.sum())
        elif ze_sum == "columns":
            reg0 += float(cen0.sum() + G.sum()); regC += cenC.sum(axis=0)
        else:
            raise ValueError(ze_sum)
        # :458-463
        t2_0 = Acl.T @ t2_0
        t2_C = Acl.T @ t2_C
        for i in range(n):
            t2_C[i, [ixb(k, c) for c in range(n)]] += deltaA[i]
            t2_C[i, [iub(k, c) for c in range(m)]] += deltaB[i]
        T1 = term1[k]
        cen0 = np.linalg.matrix_power(Acl, k + 1) @ e0 + T1.center + t2_0
        cenC = t2_C.copy()
        G = np.concatenate([np.zeros((n, 1)), T1.generators], axis=1)
        ze.append((cen0.copy(), cenC.copy(), G.copy()))
    xb_idx = np.array([[ixb(k, i) for i in range(n)] for k in range(N + 1)])
    ub_idx = np.array([[iub(k, j) for j in range(m)] for k in range(N)])
    L = loss(nxi, xb_idx[1:], ub_idx)                                                # :474
    extra = constraints(nxi, xb_idx[1:], ub_idx) if constraints else []              # :465
    ne = len(L.ab)
    nz = nxi + ne
    Am = np.zeros((len(rows) + 2 * ne + len(extra), nz))
    Am[:len(rows), :nxi] = np.array(rows)
    r_i = len(rows)
    P = np.zeros((nz, nz)); q = np.zeros(nz); r0 = reg0
    q[:nxi] += regC
    for w, F, h in L.sq:
        P[:nxi, :nxi] += 2.0 * w * F.T @ F; q[:nxi] += 2.0 * w * F.T @ h; r0 += w * float(h @ h)
    for j, (w, f, h) in enumerate(L.ab):
        q[nxi + j] = w
        for s in (+1.0, -1.0):
            Am[r_i, :nxi] = -s * f; Am[r_i, nxi + j] = 1.0
            lo.append(s * h); hi.append(np.inf); r_i += 1
    for a, l_, h_ in extra:
        Am[r_i, :nxi] = a; lo.append(l_); hi.append(h_); r_i += 1
    return dict(P=P, q=q, r=r0, A=Am, l=np.array(lo), u=np.array(hi), nxi=nxi, N=N, n=n, m=m, o_v=o_v, o_ub=o_ub, ze=ze)


def solve(*args, tol=1e-12, **kw):
    """-> dict(result, v (N, m), xbar (N+1, n), ubar (N, m), ze1 (n, 1 + 1 + gamma_W + gamma_sigma0), status, cert)."""
    qp = build(*args, **kw)
    r = solve_qp(qp["P"], qp["q"], qp["A"], qp["l"], qp["u"], tol=tol)
    N, n, m = qp["N"], qp["n"], qp["m"]
    out = dict(status=r.status, cert=r.cert, qp=qp)
    if r.status != "solved":
        out["result"] = np.inf                                                       # :496-497 'Problem is unbounded'
        return out
    xi = r.x[:qp["nxi"]]
    out["result"] = r.obj + qp["r"]
    out["xbar"] = xi[:(N + 1) * n].reshape(N + 1, n)
    out["v"] = xi[qp["o_v"]:qp["o_v"] + N * m].reshape(N, m)
    out["ubar"] = xi[qp["o_ub"]:qp["o_ub"] + N * m].reshape(N, m)
    c0, cC, G = qp["ze"][1]
    out["ze1"] = np.concatenate([(c0 + cC @ xi)[:, None], G], axis=1)               # :499  Ze[1]
    return out
