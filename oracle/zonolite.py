"""Literal zonotope algebra  --  ORACLE (test infrastructure, see oracle/__init__.py).

Restates, from call-site shapes and the CORA toolbox definitions that ``pyzonotope`` ports, the
operations the reference uses on its hot path (``tzddpc/tzddpc.py:6`` imports
``MatrixZonotope, concatenate_zonotope, Zonotope, CVXZonotope, Interval``; call sites
``:81, :119, :123, :126-128, :172-207``).  The third-party source is not in this image
(un-pinned dependency, reference ``setup.py:12``): PARITY UNPINNED.

Semantics assumed:
  Zonotope(c, G)            Z = [c | G]; set {c + G b : |b|_inf <= 1}
  Z.interval                c -/+ sum_j |G[:, j]|
  Z + vec                   shifts the center;  Z1 + Z2 Minkowski sum (centers add, generators
                            concatenate)
  Z * M   (M ndarray)       LEFT multiplication M @ Z   (required by ``tzddpc/tzddpc.py:192``:
                            Ze is n-dim, K is m x n, result must be m-dim)
  MatrixZonotope(C, {G_i})  {C + sum_i b_i G_i}
  Mz * Z                    columns [C Z, G_1 Z, ..., G_g Z] with Z = [c | G]: first column is the
                            center, the remaining (g+1)(q+1)-1 are generators
  Mz * M  (M ndarray)       <C M, {G_i M}>                         (``:119``)
  Mz + M                    shifts the center                       (``:123``)
  reduce(order)             Girard: generators sorted by |g|_1-|g|_inf, the smallest
                            (num - floor(d (order-1))) are boxed into diag(sum |g|)
  concatenate_zonotope(W,T) n x T matrix zonotope, center [c_W ... c_W], one generator per
                            (generator of W, column)               (``:81``)
  compute_LTI_matrix_zonotope(Xm, Xp, Um, Mw) = (Xp^T - Mw) pinv([Xm^T; Um^T])   (``:83``)
"""
from __future__ import annotations

import itertools
import numpy as np


class Interval:
    def __init__(self, left, right):
        self.left_limit = np.asarray(left, dtype=float)
        self.right_limit = np.asarray(right, dtype=float)

    @property
    def radius(self):
        return 0.5 * (self.right_limit - self.left_limit)

    @property
    def center(self):
        return 0.5 * (self.right_limit + self.left_limit)


class Zonotope:
    def __init__(self, center, generators):
        c = np.asarray(center, dtype=float).reshape(-1)
        G = np.asarray(generators, dtype=float)
        if G.ndim == 1:
            G = G.reshape(c.size, -1)
        assert G.shape[0] == c.size, "center / generator dimension mismatch"
        self.Z = np.hstack([c[:, None], G])

    @property
    def center(self):
        return self.Z[:, 0]

    @property
    def generators(self):
        return self.Z[:, 1:]

    @property
    def dimension(self):
        return self.Z.shape[0]

    @property
    def num_generators(self):
        return self.Z.shape[1] - 1

    @property
    def interval(self) -> Interval:
        r = np.abs(self.generators).sum(axis=1)
        return Interval(self.center - r, self.center + r)

    def __add__(self, other):
        if isinstance(other, Zonotope):
            return Zonotope(self.center + other.center, np.hstack([self.generators, other.generators]))
        return Zonotope(self.center + np.asarray(other, dtype=float).reshape(-1), self.generators)

    __radd__ = __add__

    def __mul__(self, M):
        M = np.atleast_2d(np.asarray(M, dtype=float))
        return Zonotope(M @ self.center, M @ self.generators)

    def sample(self, k: int = 1, rng=None):
        rng = np.random.default_rng() if rng is None else rng
        beta = rng.uniform(-1.0, 1.0, size=(k, self.num_generators))
        return self.center[None, :] + beta @ self.generators.T

    def compute_vertices(self):
        """All sign combinations, de-duplicated to the extreme points (small generator counts)."""
        g = self.num_generators
        pts = np.array([self.center + self.generators @ np.array(s) for s in itertools.product([-1.0, 1.0], repeat=g)])
        pts = np.unique(np.round(pts, 14), axis=0)
        if pts.shape[0] <= 2 or self.dimension == 1:
            return pts
        try:
            from scipy.spatial import ConvexHull
            return pts[ConvexHull(pts).vertices]
        except Exception:
            return pts

    def reduce(self, order: int):
        G = _girard(self.generators, order)
        return Zonotope(self.center, G)


def _girard(G: np.ndarray, order: int) -> np.ndarray:
    """Girard reduction of a generator matrix d x g to at most d*order generators."""
    d, g = G.shape
    if g <= d * order:
        return G.copy()
    h = np.abs(G).sum(axis=0) - np.abs(G).max(axis=0)
    n_unreduced = int(np.floor(d * (order - 1)))
    n_reduced = g - n_unreduced
    idx = np.argsort(h, kind="stable")
    red, unred = idx[:n_reduced], idx[n_reduced:]
    box = np.diag(np.abs(G[:, red]).sum(axis=1))
    return np.hstack([G[:, unred], box])


class MatrixZonotope:
    def __init__(self, center, generators):
        self.center = np.asarray(center, dtype=float)
        G = np.asarray(generators, dtype=float)
        if G.size == 0:
            G = np.zeros((0,) + self.center.shape)
        assert G.ndim == 3 and G.shape[1:] == self.center.shape
        self.generators = G

    @property
    def num_generators(self):
        return self.generators.shape[0]

    @property
    def shape(self):
        return self.center.shape

    def __add__(self, M):
        return MatrixZonotope(self.center + np.asarray(M, dtype=float), self.generators)

    def __mul__(self, other):
        if isinstance(other, AffZonotope):
            # [C Z, G_1 Z, ...] on the coefficient tensor (n, 1+q, 1+nv)
            parts = [np.einsum("ij,jqv->iqv", self.center, other.T)]
            parts += [np.einsum("ij,jqv->iqv", Gi, other.T) for Gi in self.generators]
            return AffZonotope(np.concatenate(parts, axis=1))
        if isinstance(other, Zonotope):
            cols = [self.center @ other.Z] + [Gi @ other.Z for Gi in self.generators]
            Z = np.hstack(cols)
            return Zonotope(Z[:, 0], Z[:, 1:])
        M = np.asarray(other, dtype=float)
        return MatrixZonotope(self.center @ M, np.einsum("gij,jk->gik", self.generators, M))

    def reduce(self, order: int):
        g = self.num_generators
        d = self.center.size
        Gv = self.generators.reshape(g, d).T            # d x g, vectorised (row-major)
        R = _girard(Gv, order)
        return MatrixZonotope(self.center, R.T.reshape((-1,) + self.center.shape))

    def sample(self, k: int = 1, rng=None):
        rng = np.random.default_rng() if rng is None else rng
        beta = rng.uniform(-1.0, 1.0, size=(k, self.num_generators))
        return self.center[None] + np.einsum("kg,gij->kij", beta, self.generators)

    def contains(self, M, tol: float = 1e-9) -> bool:
        """LP feasibility in beta in [-1,1]^g of C + sum beta_i G_i == M."""
        from scipy.optimize import linprog
        g = self.num_generators
        Aeq = self.generators.reshape(g, -1).T
        beq = (np.asarray(M, dtype=float) - self.center).reshape(-1)
        if g == 0:
            return bool(np.abs(beq).max() <= tol)
        res = linprog(np.zeros(g), A_eq=Aeq, b_eq=beq, bounds=[(-1, 1)] * g, method="highs")
        return bool(res.status == 0)


def concatenate_zonotope(W: Zonotope, T: int) -> MatrixZonotope:
    n = W.dimension
    center = np.tile(W.center[:, None], (1, T))
    gens = []
    for i in range(W.num_generators):
        for t in range(T):
            G = np.zeros((n, T))
            G[:, t] = W.generators[:, i]
            gens.append(G)
    return MatrixZonotope(center, np.array(gens).reshape(-1, n, T))


def compute_LTI_matrix_zonotope(Xm, Xp, Um, Mw: MatrixZonotope) -> MatrixZonotope:
    """(X+ - Mw) pinv([X-; U-]); data matrices are T x dim (rows = samples)."""
    D = np.vstack([np.asarray(Xm).T, np.asarray(Um).T])
    P = np.linalg.pinv(D)
    Msigma = MatrixZonotope(np.asarray(Xp).T - Mw.center, -Mw.generators)
    return Msigma * P


class AffZonotope:
    """Zonotope whose columns are affine in a decision vector (the reference's ``CVXZonotope``).

    ``T[i, j, 0]`` is the constant part of entry i of column j (column 0 = center), ``T[i, j, 1+k]``
    the coefficient of decision variable k.
    """

    def __init__(self, T):
        self.T = np.asarray(T, dtype=float)
        assert self.T.ndim == 3

    @staticmethod
    def from_center(center_aff: np.ndarray, num_zero_generators: int = 1):
        """center_aff: (n, 1+nv) affine rows; generator block of zeros as in ``:172,174``."""
        n, w = center_aff.shape
        T = np.zeros((n, 1 + num_zero_generators, w))
        T[:, 0, :] = center_aff
        return AffZonotope(T)

    @staticmethod
    def from_zonotope(Z: Zonotope, nv: int):
        T = np.zeros((Z.dimension, 1 + Z.num_generators, 1 + nv))
        T[:, :, 0] = Z.Z
        return AffZonotope(T)

    @property
    def dimension(self):
        return self.T.shape[0]

    @property
    def nv(self):
        return self.T.shape[2] - 1

    @property
    def num_generators(self):
        return self.T.shape[1] - 1

    @property
    def center(self):
        return self.T[:, 0, :]

    @property
    def generators(self):
        return self.T[:, 1:, :]

    def __add__(self, other):
        if isinstance(other, AffZonotope):
            T = np.concatenate([self.T[:, :1] + other.T[:, :1], self.T[:, 1:], other.T[:, 1:]], axis=1)
            return AffZonotope(T)
        if isinstance(other, Zonotope):
            return self + AffZonotope.from_zonotope(other, self.nv)
        # affine vector (n, 1+nv) or constant vector (n,)
        o = np.asarray(other, dtype=float)
        T = self.T.copy()
        if o.ndim == 1:
            T[:, 0, 0] += o
        else:
            T[:, 0, :] += o
        return AffZonotope(T)

    def __mul__(self, M):
        M = np.atleast_2d(np.asarray(M, dtype=float))
        return AffZonotope(np.einsum("ij,jqv->iqv", M, self.T))

    def value(self, x: np.ndarray) -> Zonotope:
        """Numeric zonotope at decision vector x."""
        Z = self.T[:, :, 0] + self.T[:, :, 1:] @ x
        return Zonotope(Z[:, 0], Z[:, 1:])
