"""cplite -- the slice of the cvxpy modelling surface that TZDDPC callbacks use.

The reference hands cvxpy ``Variable`` objects to the user's ``build_loss(u, x)`` /
``build_constraints(u, x)`` callbacks (reference ``tzddpc/tzddpc.py:213, 222, 327, 336``) and lets
cvxpy canonicalise the result.  On the MI355X path the problem is assembled once, on the host,
into a parametric QP whose per-step solve runs in HIP kernels, so the callbacks receive
look-alike objects that *record* what is done to them.  Supported (everything the reference
examples use, ``examples/1.double_integrator_sim.py:22-34``, ``examples/2.pulley_sim.py:17-28``,
``examples/3.5dimsystem_sim.py:14-26``):

  indexing / slicing, ``.shape``, ``+ - *`` with scalars / arrays, ``@`` with constant matrices,
  ``norm(e, 2) ** 2``, ``sum_squares``, ``quad_form`` (PSD), ``norm(e, 1)``, ``abs``, ``norm1``,
  ``norm(e, 'inf')``, ``norm(scalar, 2)`` (== abs), ``sum``, comparisons ``<= >= ==`` between
  affine expressions and constants.

``norm(vector, 2)`` un-squared (a second-order cone) is *recorded* (``Convex.soc``) and the builder decides: on the free
variable ``u`` of ``build_problem`` (reference ``tzddpc/tzddpc.py:160, 222``: constrained by nothing) its minimum is 0 and the
term drops out exactly -- ``1e-1 * cp.norm(u[i], p=2)`` of ``examples/3.5dimsystem_sim.py:19`` with ``dim_u > 1`` --; anywhere
else it raises ``CpliteError``.  Anything else (products of expressions, non-convex use) raises ``CpliteError`` at build time
-- never a silent fallback.

Use in an example:  ``from tzddpc_amd import cplite as cp``  instead of ``import cvxpy as cp``.
"""
from __future__ import annotations

import numbers
from typing import List, Sequence

import numpy as np


class CpliteError(Exception):
    pass


def _is_num(x):
    return isinstance(x, (numbers.Number, np.ndarray, list, tuple)) and not isinstance(x, (Affine, Convex))


class Affine:
    """Array-valued affine expression  value = C @ xi + d  over a symbol vector xi."""
    __array_priority__ = 1000

    def __init__(self, C: np.ndarray, d: np.ndarray, shape):
        self.C = C
        self.d = d
        self._shape = tuple(shape)

    # -- structure ---------------------------------------------------------------------
    @property
    def shape(self):
        return self._shape

    @property
    def size(self):
        return int(np.prod(self._shape)) if self._shape else 1

    @property
    def ndim(self):
        return len(self._shape)

    @property
    def nsym(self):
        return self.C.shape[1]

    def __len__(self):
        if not self._shape:
            raise TypeError("len() of a scalar expression")
        return self._shape[0]

    def _idx(self):
        return np.arange(self.size).reshape(self._shape)

    def __getitem__(self, key):
        sel = self._idx()[key]
        flat = np.asarray(sel).reshape(-1)
        return Affine(self.C[flat], self.d[flat], np.shape(sel))

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    @property
    def T(self):
        sel = self._idx().T
        flat = sel.reshape(-1)
        return Affine(self.C[flat], self.d[flat], sel.shape)

    def flatten(self):
        return Affine(self.C, self.d, (self.size,))

    def is_dcp(self):
        return True

    def is_constant(self):
        return not np.any(self.C)

    # -- arithmetic --------------------------------------------------------------------
    @staticmethod
    def const(val, nsym):
        a = np.asarray(val, dtype=float)
        return Affine(np.zeros((a.size, nsym)), a.reshape(-1).copy(), a.shape)

    def _coerce(self, other):
        if isinstance(other, Affine):
            return other
        if isinstance(other, Convex):
            return NotImplemented
        return Affine.const(other, self.nsym)

    def _broadcast(self, other):
        if self._shape == other._shape:
            return self, other, self._shape
        shp = np.broadcast_shapes(self._shape, other._shape)
        ia = np.broadcast_to(self._idx(), shp).reshape(-1)
        ib = np.broadcast_to(other._idx(), shp).reshape(-1)
        return Affine(self.C[ia], self.d[ia], shp), Affine(other.C[ib], other.d[ib], shp), shp

    def __add__(self, other):
        if isinstance(other, Convex):
            return other.__radd__(self)
        o = self._coerce(other)
        a, b, shp = self._broadcast(o)
        return Affine(a.C + b.C, a.d + b.d, shp)

    __radd__ = __add__

    def __neg__(self):
        return Affine(-self.C, -self.d, self._shape)

    def __sub__(self, other):
        if isinstance(other, Convex):
            raise CpliteError("affine - convex is not convex")
        return self + (-self._coerce(other))

    def __rsub__(self, other):
        return (-self) + other

    def __mul__(self, other):
        if isinstance(other, (Affine, Convex)):
            if isinstance(other, Affine) and other.is_constant():
                other = other.d.reshape(other.shape)
            elif isinstance(other, Affine) and self.is_constant():
                return other * self.d.reshape(self.shape)
            else:
                raise CpliteError("product of two expressions is not supported (not DCP-affine)")
        w = np.asarray(other, dtype=float)
        if w.ndim == 0:
            return Affine(self.C * float(w), self.d * float(w), self._shape)
        wa = Affine.const(w, self.nsym)
        a, b, shp = self._broadcast(wa)
        return Affine(a.C * b.d[:, None], a.d * b.d, shp)

    __rmul__ = __mul__

    def __truediv__(self, other):
        return self * (1.0 / np.asarray(other, dtype=float))

    def __matmul__(self, M):
        M = np.asarray(M, dtype=float)
        if self.ndim == 1:
            idx = self._idx()
            rows = M.T if M.ndim == 2 else M.reshape(1, -1)
            C = rows @ self.C[idx]; d = rows @ self.d[idx]
            return Affine(np.atleast_2d(C), np.atleast_1d(d), (M.shape[1],) if M.ndim == 2 else ())
        if self.ndim == 2:
            r, c = self._shape
            outc = M.shape[1] if M.ndim == 2 else 1
            M2 = M.reshape(c, outc)
            C = np.einsum("rcs,co->ros", self.C.reshape(r, c, -1), M2).reshape(r * outc, -1)
            d = (self.d.reshape(r, c) @ M2).reshape(-1)
            return Affine(C, d, (r, outc) if M.ndim == 2 else (r,))
        raise CpliteError("@ needs a 1-D or 2-D expression")

    def __rmatmul__(self, M):
        M = np.asarray(M, dtype=float)
        if self.ndim == 1:
            M2 = np.atleast_2d(M)
            C = M2 @ self.C; d = M2 @ self.d
            return Affine(C, d, (M2.shape[0],) if M.ndim == 2 else ())
        if self.ndim == 2:
            r, c = self._shape
            M2 = np.atleast_2d(M)
            C = np.einsum("or,rcs->ocs", M2, self.C.reshape(r, c, -1)).reshape(M2.shape[0] * c, -1)
            d = (M2 @ self.d.reshape(r, c)).reshape(-1)
            return Affine(C, d, (M2.shape[0], c) if M.ndim == 2 else (c,))
        raise CpliteError("@ needs a 1-D or 2-D expression")

    def __pow__(self, p):
        if p == 2:
            return square(self)
        raise CpliteError("only ** 2 is supported")

    # -- comparisons -> constraints ------------------------------------------------------
    def __le__(self, other):
        return Constraint(self - self._coerce(other), "<=")

    def __ge__(self, other):
        return Constraint(self - self._coerce(other), ">=")

    def __eq__(self, other):  # noqa: D105  (cvxpy overloads == the same way)
        return Constraint(self - self._coerce(other), "==")

    __hash__ = None

    def value_at(self, xi):
        return (self.C @ xi + self.d).reshape(self._shape)


Expression = Affine
Variable = Affine


class Constraint:
    """``expr (<=|>=|==) 0`` element-wise, ``expr`` affine."""

    def __init__(self, expr: Affine, kind: str):
        self.expr = expr
        self.kind = kind

    def is_dcp(self):
        return True


class Convex:
    """Scalar convex expression: const + linear + sum w||F xi + h||^2 + sum w|f'xi + h| + sum w max|.|."""

    def __init__(self, nsym):
        self.nsym = nsym
        self.const = 0.0
        self.lin = np.zeros(nsym)
        self.sq: list = []      # (w, Affine vector)
        self.ab: list = []      # (w, Affine vector)  -> w * sum |.|
        self.mx: list = []      # (w, Affine vector)  -> w * max |.|
        self.soc: list = []     # (w, Affine vector)  -> w * ||.||_2 : only the builder knows whether it can be honoured

    shape = ()

    def is_dcp(self):
        return True

    def copy(self):
        c = Convex(self.nsym)
        c.const = self.const; c.lin = self.lin.copy()
        c.sq = list(self.sq); c.ab = list(self.ab); c.mx = list(self.mx); c.soc = list(self.soc)
        return c

    def __add__(self, other):
        out = self.copy()
        if isinstance(other, Convex):
            out.const += other.const; out.lin += other.lin
            out.sq += other.sq; out.ab += other.ab; out.mx += other.mx; out.soc += other.soc
        elif isinstance(other, Affine):
            if other.size != 1:
                raise CpliteError("loss must be scalar")
            out.const += float(other.d[0]); out.lin += other.C[0]
        else:
            out.const += float(other)
        return out

    __radd__ = __add__

    def __mul__(self, w):
        if isinstance(w, (Affine, Convex)):
            raise CpliteError("product of expressions is not supported")
        w = float(w)
        if w < 0:
            raise CpliteError("negative multiple of a convex term is not convex")
        out = Convex(self.nsym)
        out.const = self.const * w; out.lin = self.lin * w
        out.sq = [(a * w, e) for a, e in self.sq]
        out.ab = [(a * w, e) for a, e in self.ab]
        out.mx = [(a * w, e) for a, e in self.mx]
        out.soc = [(a * w, e) for a, e in self.soc]
        return out

    __rmul__ = __mul__

    def __truediv__(self, w):
        return self * (1.0 / float(w))

    def __pow__(self, p):
        raise CpliteError("power of a general convex expression is not supported")

    def value_at(self, xi):
        v = self.const + self.lin @ xi
        for w, e in self.sq:
            v += w * float(np.sum(e.value_at(xi) ** 2))
        for w, e in self.ab:
            v += w * float(np.sum(np.abs(e.value_at(xi))))
        for w, e in self.mx:
            v += w * float(np.max(np.abs(e.value_at(xi))))
        for w, e in self.soc:
            v += w * float(np.sqrt(np.sum(e.value_at(xi) ** 2)))
        return v


SOC_MESSAGE = ("norm(vector, 2) (a second-order cone) is not supported by the QP path except on the free input variable of "
               "build_problem, where it vanishes; use norm(., 2)**2, norm(., 1) or norm(., 'inf')")


class _Norm2(Convex):
    """``norm(e, 2)``: squared it is a quadratic, of a scalar it is ``abs``; of a vector it is a second-order cone, recorded in
    ``soc`` for the builder to drop (free ``u``) or refuse (``SOC_MESSAGE``)."""

    def __init__(self, e: Affine):
        super().__init__(e.nsym)
        self._e = e.flatten()
        if self._e.size == 1:
            self.ab = [(1.0, self._e)]
        else:
            self.soc = [(1.0, self._e)]

    def __pow__(self, p):
        if p != 2:
            raise CpliteError("only norm(., 2) ** 2 is supported")
        out = Convex(self.nsym)
        out.sq = [(1.0, self._e)]
        return out


def _aff(e, like=None):
    if isinstance(e, Affine):
        return e
    raise CpliteError(f"expected an affine expression, got {type(e).__name__}")


def norm(e, p=2):
    e = _aff(e)
    if p == 2 or p == "fro":
        return _Norm2(e)
    out = Convex(e.nsym)
    if p == 1:
        out.ab = [(1.0, e.flatten())]
    elif p in ("inf", np.inf, float("inf")):
        out.mx = [(1.0, e.flatten())]
    else:
        raise CpliteError(f"norm with p={p!r} is not supported")
    return out


def norm1(e):
    return norm(e, 1)


def norm_inf(e):
    return norm(e, "inf")


def abs(e):  # noqa: A001  (mirrors cvxpy.abs)
    e = _aff(e)
    if e.size != 1:
        raise CpliteError("abs() of a non-scalar would be array-valued; use norm(., 1) or sum of scalars")
    return norm(e, 1)


def square(e):
    e = _aff(e)
    if e.size != 1:
        raise CpliteError("square() of a non-scalar would be array-valued; use sum_squares")
    return sum_squares(e)


def sum_squares(e):
    e = _aff(e)
    out = Convex(e.nsym)
    out.sq = [(1.0, e.flatten())]
    return out


def quad_form(e, Q):
    e = _aff(e).flatten()
    Q = np.asarray(Q, dtype=float)
    Qs = 0.5 * (Q + Q.T)
    w, V = np.linalg.eigh(Qs)
    if w.min() < -1e-10 * max(1.0, w.max()):
        raise CpliteError("quad_form needs a PSD matrix")
    L = (V * np.sqrt(np.maximum(w, 0.0))).T         # Q = L'L
    out = Convex(e.nsym)
    out.sq = [(1.0, L @ e)]
    return out


def sum(e):  # noqa: A001
    if isinstance(e, Convex):
        return e
    e = _aff(e)
    return Affine(e.C.sum(axis=0, keepdims=True), np.array([e.d.sum()]), ())


def hstack(parts: Sequence[Affine]):
    parts = [p if p.ndim else p.flatten() for p in parts]
    return Affine(np.vstack([p.C for p in parts]), np.concatenate([p.d for p in parts]), (int(np.sum([p.size for p in parts])),))


def as_convex(e, nsym) -> Convex:
    """Normalise whatever a loss callback returned into a ``Convex``."""
    if isinstance(e, Convex):
        return e
    if isinstance(e, Affine):
        return Convex(nsym) + e
    if isinstance(e, numbers.Number):
        return Convex(nsym) + float(e)
    raise CpliteError("Loss function is not defined or is not convex!")
