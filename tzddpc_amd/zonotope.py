"""Host-side set objects of the TZDDPC API surface (numpy; build-time only).

The reference takes these from the un-vendored ``pyzonotope`` / ``pydatadrivenreachability``
packages (reference ``tzddpc/tzddpc.py:6-7``; ``examples/*.py`` construct ``Zonotope(center, G)``
and read ``.sample``, ``.compute_vertices``, ``.interval``).  Only what the hot path and the
examples touch is provided.  None of this runs per MPC step: per-step tube propagation is the
HIP kernel ``tz_prepare`` (``csrc/tz_kernels.hip``), which consumes the constants extracted here.

Semantics (CORA conventions): ``Z * M`` is the linear map ``M @ Z``; ``Z1 + Z2`` the Minkowski
sum; ``MatrixZonotope * Zonotope`` stacks ``[C Z, G_1 Z, ...]``; ``reduce(1)`` is Girard's
order-1 boxing.
"""
from __future__ import annotations

import itertools
from typing import Optional

import numpy as np


class Interval:
    def __init__(self, left_limit, right_limit):
        self.left_limit = np.asarray(left_limit, dtype=float)
        self.right_limit = np.asarray(right_limit, dtype=float)

    def __repr__(self):
        return f"Interval({self.left_limit}, {self.right_limit})"


class Zonotope:
    """{ c + G beta : ||beta||_inf <= 1 };  ``Z`` attribute is ``[c | G]`` like pyzonotope's."""

    def __init__(self, center, generators):
        c = np.asarray(center, dtype=float).reshape(-1)
        G = np.asarray(generators, dtype=float)
        G = G.reshape(c.size, -1) if G.ndim != 2 else G
        if G.shape[0] != c.size:
            raise ValueError("center and generator matrix have different dimensions")
        self.Z = np.concatenate([c[:, None], G], axis=1)

    center = property(lambda self: self.Z[:, 0])
    generators = property(lambda self: self.Z[:, 1:])
    dimension = property(lambda self: self.Z.shape[0])
    num_generators = property(lambda self: self.Z.shape[1] - 1)

    @property
    def order(self):
        return self.num_generators / self.dimension

    @property
    def interval(self) -> Interval:
        rad = np.abs(self.generators).sum(axis=1)
        return Interval(self.center - rad, self.center + rad)

    def __add__(self, other):
        if isinstance(other, Zonotope):
            return Zonotope(self.center + other.center, np.concatenate([self.generators, other.generators], axis=1))
        return Zonotope(self.center + np.asarray(other, dtype=float).reshape(-1), self.generators)

    __radd__ = __add__

    def __mul__(self, M):
        M = np.atleast_2d(np.asarray(M, dtype=float))
        return Zonotope(M @ self.center, M @ self.generators)

    def sample(self, batch_size: int = 1, rng: Optional[np.random.Generator] = None) -> np.ndarray:
        """``batch_size`` points c + G beta, beta ~ U(-1, 1); global numpy RNG unless ``rng`` is given."""
        g = self.num_generators
        beta = (np.random.uniform(-1.0, 1.0, size=(batch_size, g)) if rng is None
                else rng.uniform(-1.0, 1.0, size=(batch_size, g)))
        return self.center[None, :] + beta @ self.generators.T

    def compute_vertices(self) -> np.ndarray:
        g = self.num_generators
        if g > 16:
            raise ValueError("compute_vertices enumerates 2^g sign patterns; too many generators")
        signs = np.array(list(itertools.product((-1.0, 1.0), repeat=g))).reshape(-1, g)
        pts = np.unique(np.round(self.center[None, :] + signs @ self.generators.T, 14), axis=0)
        if self.dimension >= 2 and pts.shape[0] > self.dimension + 1:
            try:
                from scipy.spatial import ConvexHull
                pts = pts[ConvexHull(pts).vertices]
            except Exception:
                pass
        return pts

    def reduce(self, order: int) -> "Zonotope":
        return Zonotope(self.center, girard_box(self.generators, order))

    def polygon_vertices(self) -> np.ndarray:
        """Boundary of a 2-dimensional zonotope, counter-clockwise, (2 g', 2) with g' the non-zero generators: the generators turned into
        the upper half-plane and sorted by angle are the edges of one half of the boundary (any number of generators, no 2^g enumeration)."""
        if self.dimension != 2:
            raise ValueError("polygon: the zonotope must be 2-dimensional")
        G = self.generators[:, np.any(self.generators != 0.0, axis=0)]
        if G.shape[1] == 0:
            return self.center[None, :].copy()
        G = np.where((G[1] < 0) | ((G[1] == 0) & (G[0] < 0)), -G, G)
        G = G[:, np.argsort(np.arctan2(G[1], G[0]), kind="stable")]
        start = self.center - G.sum(axis=1)                      # the lowest vertex
        half = start[None, :] + 2.0 * np.cumsum(G, axis=1).T     # ... up the right side to the highest
        return np.concatenate([start[None, :], half[:-1], 2.0 * self.center[None, :] - start[None, :], 2.0 * self.center[None, :] - half[:-1]], axis=0)

    @property
    def polygon(self):
        """What the reference's plotting code draws (``examples/1.double_integrator_sim.py:170,174``: ``Z.reduce(3).polygon`` in a
        PatchCollection): a ``matplotlib.patches.Polygon`` when matplotlib is installed, the vertex array otherwise."""
        V = self.polygon_vertices()
        try:
            from matplotlib.patches import Polygon
        except ImportError:
            return V
        return Polygon(V, closed=True)

    def __repr__(self):
        return f"Zonotope(dim={self.dimension}, generators={self.num_generators})"


def girard_box(G: np.ndarray, order: int) -> np.ndarray:
    """Girard reduction: keep the floor(d(order-1)) 'longest' generators, box the rest."""
    d, g = G.shape
    if g <= d * order:
        return G.copy()
    a = np.abs(G)
    score = a.sum(axis=0) - a.max(axis=0)
    keep = int(np.floor(d * (order - 1)))
    order_idx = np.argsort(score, kind="stable")
    boxed, kept = order_idx[:g - keep], order_idx[g - keep:]
    return np.concatenate([G[:, kept], np.diag(a[:, boxed].sum(axis=1))], axis=1)


class MatrixZonotope:
    """{ C + sum_i beta_i G_i }, generators stored as an array (g, rows, cols)."""

    def __init__(self, center, generators):
        self.center = np.asarray(center, dtype=float)
        G = np.asarray(generators, dtype=float)
        self.generators = G.reshape((-1,) + self.center.shape)

    num_generators = property(lambda self: self.generators.shape[0])
    shape = property(lambda self: self.center.shape)

    def __add__(self, M):
        return MatrixZonotope(self.center + np.asarray(M, dtype=float), self.generators)

    def __mul__(self, other):
        if isinstance(other, Zonotope):
            blocks = [self.center @ other.Z] + [G @ other.Z for G in self.generators]
            Z = np.concatenate(blocks, axis=1)
            return Zonotope(Z[:, 0], Z[:, 1:])
        M = np.asarray(other, dtype=float)
        return MatrixZonotope(self.center @ M, self.generators @ M)

    def reduce(self, order: int) -> "MatrixZonotope":
        g = self.num_generators
        flat = self.generators.reshape(g, -1).T
        R = girard_box(flat, order)
        return MatrixZonotope(self.center, R.T.reshape((-1,) + self.center.shape))

    def sample(self, batch_size: int = 1, rng: Optional[np.random.Generator] = None) -> np.ndarray:
        g = self.num_generators
        beta = (np.random.uniform(-1.0, 1.0, size=(batch_size, g)) if rng is None
                else rng.uniform(-1.0, 1.0, size=(batch_size, g)))
        return self.center[None] + np.tensordot(beta, self.generators, axes=(1, 0))

    def contains(self, M, tol: float = 1e-9) -> bool:
        from scipy.optimize import linprog
        g = self.num_generators
        rhs = (np.asarray(M, dtype=float) - self.center).reshape(-1)
        if g == 0:
            return bool(np.abs(rhs).max(initial=0.0) <= tol)
        res = linprog(np.zeros(g), A_eq=self.generators.reshape(g, -1).T, b_eq=rhs,
                      bounds=[(-1.0, 1.0)] * g, method="highs")
        return res.status == 0

    def single_entry_magnitudes(self) -> Optional[np.ndarray]:
        """|.|-sum of the generators if every generator has exactly <= 1 non-zero entry, else None.

        Exact structure test (counts non-zeros) that gates the collapsed tube path.
        """
        nnz = np.count_nonzero(self.generators.reshape(self.num_generators, -1), axis=1)
        if np.any(nnz > 1):
            return None
        return np.abs(self.generators).sum(axis=0)

    def __repr__(self):
        return f"MatrixZonotope(shape={self.shape}, generators={self.num_generators})"


def boxed_matrix_zonotope(center, magnitudes) -> MatrixZonotope:
    """The matrix zonotope ``reduce(1)`` returns when it boxes everything (Girard order 1, reference ``tzddpc/tzddpc.py:126-128``):
    one single-entry generator per matrix entry (row-major), magnitude ``magnitudes[r, c]``."""
    center = np.asarray(center, dtype=float)
    v = np.asarray(magnitudes, dtype=float).reshape(-1)
    return MatrixZonotope(center, np.diag(v).reshape((-1,) + center.shape))


def concatenate_zonotope(W: Zonotope, num_columns: int) -> MatrixZonotope:
    """n x T matrix zonotope of T independent copies of W (reference ``tzddpc/tzddpc.py:81``)."""
    n, g = W.dimension, W.num_generators
    center = np.repeat(W.center[:, None], num_columns, axis=1)
    gens = np.zeros((g, num_columns, n, num_columns))
    t = np.arange(num_columns)
    for i in range(g):
        gens[i, t, :, t] = W.generators[:, i]
    return MatrixZonotope(center, gens.reshape(g * num_columns, n, num_columns))


def compute_LTI_matrix_zonotope(Xm, Xp, Um, Mw: MatrixZonotope) -> MatrixZonotope:
    """Set of [A B] consistent with the data: (X+ - Mw) pinv([X-; U-])  (reference ``:83``)."""
    D = np.concatenate([np.asarray(Xm, dtype=float).T, np.asarray(Um, dtype=float).T], axis=0)
    return MatrixZonotope(np.asarray(Xp, dtype=float).T - Mw.center, -Mw.generators) * np.linalg.pinv(D)
