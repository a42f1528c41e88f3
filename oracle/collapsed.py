"""Collapsed (exactly aggregated) restatement of the TZDDPC problem -- ORACLE, any horizon.

Test infrastructure only (see oracle/__init__.py); PARITY UNPINNED.

Same control flow as reference ``tzddpc/tzddpc.py:172-207`` / ``:283-324`` (the loops are
repeated literally), but every zonotope is carried as

    Z = < c , D , [(P_l, b_l)] >     set  c + D beta + sum_l P_l diag(b_l) beta_l

with numeric center ``c``, numeric dense generators ``D`` and *layers*: a numeric matrix ``P_l``
times an axis-aligned box whose half-widths ``b_l >= 0`` are affine in t_j = |[xbar_j; v_j]|.
When every generator of ``MdataK`` / ``Mdelta`` has a single non-zero entry (what the Girard
order-1 ``reduce(1)`` at ``:126-128`` produces) the literal product satisfies, generator for
generator up to merging parallel generators of identical scalar factor,

    M_K * Z = < C_K c , C_K D , [(C_K P_l, b_l)] + [(I, Delta_K (|c| + rowabs(Z)))] >
    rowabs(Z) = sum_j |D[:, j]| + sum_l |P_l| b_l            (interval-hull radius, ``:191``)

which keeps the size linear in the horizon.  ``tests/test_oracle_collapse.py`` checks this
against ``oracle.literal`` (literal stacking) to 1e-12 for N <= 4.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np

from .zonolite import MatrixZonotope, Zonotope


def single_entry_abs(M: MatrixZonotope) -> np.ndarray:
    """Delta[r, c] = sum_i |G_i[r, c]|; asserts the single-non-zero structure."""
    for G in M.generators:
        assert np.count_nonzero(G) <= 1, "collapse needs single-entry generators (reduce(1) output)"
    return np.abs(M.generators).sum(axis=0) if M.num_generators else np.zeros(M.shape)


class AffB:
    """b = b0 + Bt @ t  (t = stacked |zeta_j|, nonnegative coefficients)."""
    __slots__ = ("b0", "Bt")

    def __init__(self, b0, Bt):
        self.b0 = b0; self.Bt = Bt

    def __add__(self, o):
        return AffB(self.b0 + o.b0, self.Bt + o.Bt)

    def lmul(self, M):
        return AffB(M @ self.b0, M @ self.Bt)


@dataclass
class CZ:
    c: np.ndarray
    D: np.ndarray
    layers: list     # [(P, AffB)]

    def rowabs(self, L=None) -> AffB:
        """radius of L @ Z (L = identity by default)."""
        n = self.c.size
        L = np.eye(n) if L is None else L
        nt = self.layers[0][1].Bt.shape[1] if self.layers else 0
        out = AffB(np.abs(L @ self.D).sum(axis=1), np.zeros((L.shape[0], nt)))
        for P, b in self.layers:
            out = out + b.lmul(np.abs(L @ P))
        return out


def mk_times(CK, DK, Z: CZ, nt: int) -> CZ:
    n = CK.shape[0]
    ra = Z.rowabs()
    if ra.Bt.shape[1] == 0:
        ra = AffB(ra.b0, np.zeros((n, nt)))
    newb = AffB(DK @ (np.abs(Z.c) + ra.b0), DK @ ra.Bt)
    layers = [(CK @ P, b) for P, b in Z.layers] + [(np.eye(n), newb)]
    return CZ(CK @ Z.c, CK @ Z.D, layers)


def cz_add(Z1: CZ, Z2: CZ) -> CZ:
    return CZ(Z1.c + Z2.c, np.hstack([Z1.D, Z2.D]), Z1.layers + Z2.layers)


def build_collapsed(A, B, MdataK: MatrixZonotope, Mdelta: MatrixZonotope, K, W: Zonotope, X: Zonotope, U: Zonotope,
                    N: int, e0, xbar0, loss: Callable, constraints: Optional[Callable] = None,
                    k0: Optional[int] = None):
    """Returns QP dict over z = [xbar | v | t | loss-epigraphs] (same xi indexing as literal)."""
    A = np.asarray(A, float); B = np.asarray(B, float); K = np.atleast_2d(np.asarray(K, float))
    n, m = B.shape
    p = n + m
    nxi = (N + 1) * n + N * m
    nt = N * p
    ixb = lambda k, i: k * n + i
    ixv = lambda k, j: (N + 1) * n + k * m + j
    e0 = np.asarray(e0, float).reshape(n); xbar0 = np.asarray(xbar0, float).reshape(n)
    CK = MdataK.center
    DK = single_entry_abs(MdataK)
    Dd = single_entry_abs(Mdelta)
    assert np.abs(Mdelta.center).max() == 0.0, "Mdelta must have zero center (reference :122-123)"

    Ze = [CZ(e0.copy(), np.zeros((n, 0)), [])]                                     # :172
    term1 = [mk_times(CK, DK, Ze[0], nt)]                                          # :175
    Z_noise = []
    for k in range(N):                                                             # :176
        Bt = np.zeros((n, nt)); Bt[:, k * p:(k + 1) * p] = Dd
        Z_noise.append(CZ(W.center.copy(), W.generators.copy(), [(np.eye(n), AffB(np.zeros(n), Bt))]))
    term2 = []
    for k in range(N):
        if k0 is None:
            term1.append(mk_times(CK, DK, term1[-1], nt))                          # :181
            noise = Z_noise[0]
            for j in range(1, k):                                                  # :184
                noise = cz_add(mk_times(CK, DK, noise, nt), Z_noise[j])            # :185
        else:
            term1.append(term1[-1] if k > k0 else mk_times(CK, DK, term1[-1], nt))  # :292-295
            start = max(0, k - k0)
            noise = Z_noise[start]
            for j in range(1, min(k, k0)):
                noise = cz_add(mk_times(CK, DK, noise, nt), Z_noise[start + j])
        term2.append(noise)

    Xi, Ui = X.interval, U.interval
    nz0 = nxi + nt
    rowsA, lo, hi = [], [], []

    def add(r, l_, h_):
        rowsA.append(r); lo.append(l_); hi.append(h_)

    # dynamics
    for i in range(n):
        r = np.zeros(nz0); r[ixb(0, i)] = 1.0; add(r, xbar0[i], xbar0[i])
    for k in range(N):
        for i in range(n):
            r = np.zeros(nz0); r[ixb(k + 1, i)] = 1.0
            r[[ixb(k, c) for c in range(n)]] -= A[i]
            r[[ixv(k, c) for c in range(m)]] -= B[i]
            add(r, 0.0, 0.0)
    tubes = []
    for k in range(N):
        Z = Ze[-1]
        rx = Z.rowabs() if Z.layers or Z.D.size else AffB(np.zeros(n), np.zeros((n, nt)))
        ru = Z.rowabs(K) if Z.layers or Z.D.size else AffB(np.zeros(m), np.zeros((m, nt)))
        if rx.Bt.shape[1] == 0:
            rx = AffB(rx.b0, np.zeros((n, nt))); ru = AffB(ru.b0, np.zeros((m, nt)))
        tubes.append((Z.c.copy(), rx, ru))
        for i in range(n):                                                         # :194-195
            r = np.zeros(nz0); r[ixb(k, i)] = 1.0; r[nxi:] = rx.Bt[i]
            add(r, -np.inf, Xi.right_limit[i] - Z.c[i] - rx.b0[i])
            r = np.zeros(nz0); r[ixb(k, i)] = 1.0; r[nxi:] = -rx.Bt[i]
            add(r, Xi.left_limit[i] - Z.c[i] + rx.b0[i], np.inf)
        Kc = K @ Z.c
        for j in range(m):                                                         # :196-197
            r = np.zeros(nz0); r[ixv(k, j)] = 1.0; r[nxi:] = ru.Bt[j]
            add(r, -np.inf, Ui.right_limit[j] - Kc[j] - ru.b0[j])
            r = np.zeros(nz0); r[ixv(k, j)] = 1.0; r[nxi:] = -ru.Bt[j]
            add(r, Ui.left_limit[j] - Kc[j] + ru.b0[j], np.inf)
        if k < N - 1:
            Ze.append(cz_add(term1[k], term2[k]))                                  # :205-207
    # t_j >= |zeta_j|
    for k in range(N):
        idx = [ixb(k, i) for i in range(n)] + [ixv(k, j) for j in range(m)]
        for c, ix in enumerate(idx):
            for s in (+1.0, -1.0):
                r = np.zeros(nz0); r[nxi + k * p + c] = 1.0; r[ix] = -s
                add(r, 0.0, np.inf)

    xb_idx = np.array([[ixb(k, i) for i in range(n)] for k in range(N + 1)])
    v_idx = np.array([[ixv(k, j) for j in range(m)] for k in range(N)])
    if k0 is None:
        L = loss(nxi, xb_idx, None)
        extra = constraints(nxi, xb_idx, v_idx) if constraints else []
    else:
        L = loss(nxi, xb_idx[1:], v_idx)
        extra = constraints(nxi, xb_idx[1:], v_idx) if constraints else []
    ne = len(L.ab)
    nz = nz0 + ne
    Amat = np.zeros((len(rowsA) + 2 * ne + len(extra), nz))
    Amat[:len(rowsA), :nz0] = np.array(rowsA)
    r_i = len(rowsA)
    P = np.zeros((nz, nz)); q = np.zeros(nz); r0 = 0.0
    for w, F, h in L.sq:
        P[:nxi, :nxi] += 2.0 * w * F.T @ F
        q[:nxi] += 2.0 * w * F.T @ h
        r0 += w * float(h @ h)
    for j, (w, f, h) in enumerate(L.ab):
        q[nz0 + j] = w
        for s in (+1.0, -1.0):
            Amat[r_i, :nxi] = -s * f; Amat[r_i, nz0 + j] = 1.0
            lo.append(s * h); hi.append(np.inf); r_i += 1
    for a, l_, h_ in extra:
        Amat[r_i, :nxi] = a; lo.append(l_); hi.append(h_); r_i += 1
    # e0-only part of every tube (what the device computes per step from e0): Ze_0 = <e0>, Ze_k >= 1: term1[k-1] = M_K^p <e0>
    e0tube = [(e0.copy(), np.zeros(n), np.zeros(m))]
    for k in range(1, N):
        Z1 = term1[k - 1]
        e0tube.append((Z1.c.copy(), Z1.rowabs().b0.copy(), Z1.rowabs(K).b0.copy()))
    # first tube row (dynamics rows come first: (N+1) n equalities), then per step k: n x (ub, lb), m x (ub, lb)
    return dict(P=P, q=q, r=r0, A=Amat, l=np.array(lo), u=np.array(hi), nxi=nxi, N=N, n=n, m=m, tubes=tubes,
                e0tube=e0tube, tube_row0=(N + 1) * n)


def collapsed_radii(qp, xi):
    """(center, rad_X, rad_U) per step at decision vector xi (t = |zeta| exactly)."""
    N, n, m = qp["N"], qp["n"], qp["m"]
    p = n + m
    xb = xi[:(N + 1) * n].reshape(N + 1, n); v = xi[(N + 1) * n:].reshape(N, m)
    t = np.abs(np.hstack([xb[:N], v])).reshape(-1)
    return [(c, rx.b0 + rx.Bt @ t, ru.b0 + ru.Bt @ t) for c, rx, ru in qp["tubes"]]


def extract(qp, z):
    N, n, m = qp["N"], qp["n"], qp["m"]
    xb = z[:(N + 1) * n].reshape(N + 1, n); v = z[(N + 1) * n:(N + 1) * n + N * m].reshape(N, m)
    return v, xb
