"""Literal restatement of ``TZDDPC.build_problem`` / ``build_problem_simplified`` -- ORACLE.

Test infrastructure only (see oracle/__init__.py); PARITY UNPINNED (no reference fixtures exist).

Follows reference ``tzddpc/tzddpc.py:132-241`` (full) and ``:243-355`` (simplified) line by line,
with *literal generator stacking*: every ``MatrixZonotope * CVXZonotope`` product multiplies the
generator count by (gamma+1), exactly like the reference, so this is only usable for small
horizons (the reference's own Table I stops at N=5).

Decision vector  xi = [ xbar (N+1) x n  |  v  N x m ]  (row-major), parameters e0, xbar0 are
numeric at build time.  The free variables ``x`` and ``u`` of the reference (``:159-160``) are
handled as follows: ``x[k] == center(Ze_k)`` is definitional and dropped; ``u`` is constrained by
nothing (``:222``), so the loss callback receives a separate free block.

The interval hull ``c -/+ sum_j |g_j|`` (``:191-197``) of an affine-in-xi generator matrix is made
an LP/QP by one epigraph variable per non-constant generator entry (this is what cvxpy's ``abs``
canonicalisation does).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

from .zonolite import AffZonotope, MatrixZonotope, Zonotope


@dataclass
class AffineLoss:
    """sum_i w_i ||F_i xi + h_i||^2  +  sum_j w_j |f_j' xi + h_j|   (all the example losses)."""
    sq: list = field(default_factory=list)    # (w, F (r x nxi), h (r))
    ab: list = field(default_factory=list)    # (w, f (nxi), h)


@dataclass
class LiteralProblem:
    N: int
    n: int
    m: int
    nxi: int
    Ze: List[AffZonotope]
    eq_A: np.ndarray
    eq_b: np.ndarray
    # box rows on affine-with-abs expressions:  a'xi + c + s * sum_j |G_j xi + g_j|  (<= ub | >= lb)
    rows: list
    loss: AffineLoss
    extra: list                                # user rows (a, lo, hi)

    def ix_xbar(self, k, i):
        return k * self.n + i

    def ix_v(self, k, j):
        return (self.N + 1) * self.n + k * self.m + j


def _aff_rows(nxi, idx):
    """(len(idx), 1+nxi) affine selector rows."""
    M = np.zeros((len(idx), 1 + nxi))
    for r, k in enumerate(idx):
        M[r, 1 + k] = 1.0
    return M


def build_literal(A, B, MdataK: MatrixZonotope, Mdelta: MatrixZonotope, K, W: Zonotope, X: Zonotope, U: Zonotope,
                  N: int, e0, xbar0, loss: Callable, constraints: Optional[Callable] = None,
                  k0: Optional[int] = None) -> LiteralProblem:
    """loss(nxi, xbar_index, u_index) -> AffineLoss ;  constraints(...) -> [(a, lo, hi)].

    ``k0 is None``: ``build_problem`` (``:172-207``); else ``build_problem_simplified`` (``:283-324``).
    """
    A = np.asarray(A, float); B = np.asarray(B, float); K = np.atleast_2d(np.asarray(K, float))
    n, m = B.shape
    nxi = (N + 1) * n + N * m
    ixb = lambda k, i: k * n + i
    ixv = lambda k, j: (N + 1) * n + k * m + j
    e0 = np.asarray(e0, float).reshape(n); xbar0 = np.asarray(xbar0, float).reshape(n)

    # dynamics  (:166-170 / :277-281)
    eqA, eqb = [], []
    for i in range(n):
        r = np.zeros(nxi); r[ixb(0, i)] = 1.0
        eqA.append(r); eqb.append(xbar0[i])
    for k in range(N):
        for i in range(n):
            r = np.zeros(nxi); r[ixb(k + 1, i)] = 1.0
            for c in range(n):
                r[ixb(k, c)] -= A[i, c]
            for c in range(m):
                r[ixv(k, c)] -= B[i, c]
            eqA.append(r); eqb.append(0.0)

    # Ze[0] = <e0, [0]>   (:172)
    c0 = np.zeros((n, 1 + nxi)); c0[:, 0] = e0
    Ze = [AffZonotope.from_center(c0, 1)]
    # XU[k] = <[xbar_k; v_k], [0]>   (:174)
    XU = []
    for k in range(N):
        XU.append(AffZonotope.from_center(_aff_rows(nxi, [ixb(k, i) for i in range(n)] + [ixv(k, j) for j in range(m)]), 1))
    term1 = [MdataK * Ze[0]]                                   # :175
    Z_noise = [Mdelta * XU[k] + W for k in range(N)]           # :176
    term2 = []
    for k in range(N):
        if k0 is None:
            term1.append(MdataK * term1[-1])                   # :181
            noise = Z_noise[0]                                 # :183
            for j in range(1, k):                              # :184  (range(1,k): reproduce, do not fix)
                noise = MdataK * noise + Z_noise[j]            # :185
        else:
            term1.append(term1[-1] if k > k0 else MdataK * term1[-1])     # :292-295
            start = max(0, k - k0)                             # :297
            noise = Z_noise[start]                             # :298
            for j in range(1, min(k, k0)):                     # :299
                noise = MdataK * noise + Z_noise[start + j]    # :300
        term2.append(noise)

    Xi, Ui = X.interval, U.interval
    rows = []
    for k in range(N):
        Zcur = Ze[-1]
        xk = _aff_rows(nxi, [ixb(k, i) for i in range(n)])
        vk = _aff_rows(nxi, [ixv(k, j) for j in range(m)])
        Zx = Zcur + xk                                         # :191
        Zu = Zcur * K + vk                                     # :192
        for i in range(n):
            rows.append(("ub", Zx.center[i], Zx.generators[i], +1.0, Xi.right_limit[i]))   # :194
            rows.append(("lb", Zx.center[i], Zx.generators[i], -1.0, Xi.left_limit[i]))    # :195
        for j in range(m):
            rows.append(("ub", Zu.center[j], Zu.generators[j], +1.0, Ui.right_limit[j]))   # :196
            rows.append(("lb", Zu.center[j], Zu.generators[j], -1.0, Ui.left_limit[j]))    # :197
        if k < N - 1:   # Ze[N] is built by the reference (:205-207) but never constrained
            Ze.append(term1[k] + term2[k])

    prob = LiteralProblem(N, n, m, nxi, Ze, np.array(eqA), np.array(eqb), rows, None, [])
    xb_idx = np.array([[ixb(k, i) for i in range(n)] for k in range(N + 1)])
    v_idx = np.array([[ixv(k, j) for j in range(m)] for k in range(N)])
    if k0 is None:
        # loss(u, xbar) with u FREE (:222); constraints(v, xbar) (:213)
        prob.loss = loss(nxi, xb_idx, None)
        prob.extra = constraints(nxi, xb_idx, v_idx) if constraints else []
    else:
        # loss(v, xbar[1:]) (:336); constraints(v, xbar[1:]) (:327)
        prob.loss = loss(nxi, xb_idx[1:], v_idx)
        prob.extra = constraints(nxi, xb_idx[1:], v_idx) if constraints else []
    return prob


def to_qp(prob: LiteralProblem, tol_const: float = 0.0):
    """Assemble  min 1/2 z'Pz + q'z + r  s.t.  l <= A z <= u  with z = [xi | epigraph vars]."""
    nxi = prob.nxi
    epi = []          # list of affine forms (1+nxi) that need an epigraph variable
    epi_key = {}

    def epi_var(form):
        key = np.round(form / (np.abs(form).max() + 1e-300), 12).tobytes() + np.float64(np.abs(form).max()).tobytes()
        if key not in epi_key:
            epi_key[key] = len(epi)
            epi.append(form)
        return epi_key[key]

    row_specs = []
    for kind, cen, gens, sgn, bound in prob.rows:
        const = cen[0]
        lin = cen[1:].copy()
        tcoef = {}
        for g in gens:
            if np.abs(g[1:]).max(initial=0.0) <= tol_const:
                const += sgn * abs(g[0])
            else:
                j = epi_var(g)
                tcoef[j] = tcoef.get(j, 0.0) + 1.0
        row_specs.append((kind, lin, tcoef, sgn, bound - const))
    loss_epi = []
    for w, f, h in prob.loss.ab:
        loss_epi.append((w, epi_var(np.concatenate([[h], f]))))
    ne = len(epi)
    nz = nxi + ne
    rowsA, lo, hi = [], [], []
    for a, b in zip(prob.eq_A, prob.eq_b):
        r = np.zeros(nz); r[:nxi] = a
        rowsA.append(r); lo.append(b); hi.append(b)
    for kind, lin, tcoef, sgn, rhs in row_specs:
        r = np.zeros(nz); r[:nxi] = lin
        for j, c in tcoef.items():
            r[nxi + j] += sgn * c
        rowsA.append(r)
        if kind == "ub":
            lo.append(-np.inf); hi.append(rhs)
        else:
            lo.append(rhs); hi.append(np.inf)
    for j, form in enumerate(epi):           # t_j >= +-(form)
        for s in (+1.0, -1.0):
            r = np.zeros(nz); r[:nxi] = -s * form[1:]; r[nxi + j] = 1.0
            rowsA.append(r); lo.append(s * form[0]); hi.append(np.inf)
    for a, l_, h_ in prob.extra:
        r = np.zeros(nz); r[:nxi] = a
        rowsA.append(r); lo.append(l_); hi.append(h_)
    P = np.zeros((nz, nz)); q = np.zeros(nz); r0 = 0.0
    for w, F, h in prob.loss.sq:
        P[:nxi, :nxi] += 2.0 * w * F.T @ F
        q[:nxi] += 2.0 * w * F.T @ h
        r0 += w * float(h @ h)
    for w, j in loss_epi:
        q[nxi + j] += w
    return dict(P=P, q=q, r=r0, A=np.array(rowsA), l=np.array(lo), u=np.array(hi), nxi=nxi, ne=ne)


def literal_radii(prob: LiteralProblem, xi: np.ndarray):
    """Numeric (center, rad_X, rad_U is done by caller) of every Ze[k] at decision vector xi."""
    out = []
    for Z in prob.Ze:
        Zn = Z.value(xi)
        out.append((Zn.center.copy(), np.abs(Zn.generators).sum(axis=1)))
    return out
