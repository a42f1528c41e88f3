"""Literal stacked-generator form of the tubes ``Ze[k]`` -- host side of kernel K1g (``tz_genstack_*``).

The reference builds ``Ze[k]`` by literal zonotope algebra (``tzddpc/tzddpc.py:172-207`` full, ``:283-324`` simplified):
``MatrixZonotope * CVXZonotope`` stacks ``[C Z, G_1 Z, ..., G_gamma Z]`` (every product multiplies the generator count by
gamma + 1), ``+`` concatenates generators.  Whatever the generators of ``MdataK`` / ``Mdelta`` look like (single-entry boxes
or dense), every generator column of every ``Ze[k]`` is an affine function of ONE source vector,

    g = m0 + M xi_src ,      xi_src  in  { e0 ,  zeta_j = [xbar_j ; v_j]  (j < N) }    or constant (src = -1),

because the e0 chain (``term1``) and the noise chain (``term2``) are propagated separately and only Minkowski-summed
(``:205-207``), and ``Mdelta`` has a zero centre (``:122-123``).  This module assembles that stack once per problem (numpy,
build time, generator for generator in the reference's order); the device then evaluates, for a batch of trajectories,

    centre_k,   rad^x_k = sum_g |g|,   rad^u_k = sum_g |K g|          (the ``.interval`` of ``:191-192``)

or the generator columns themselves (the ``Ze[1]`` the reference returns from ``solve``, ``:377``) by streaming the stack from
HBM (``tzddpc_amd/csrc/tz_genstack.hip.h``).  Nothing here runs per MPC step.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

SRC_NONE, SRC_E0 = -1, 0          # source ids: -1 constant, 0 e0, 1 + j: zeta_j


class _Gen:
    """One generator (or a centre): m0 + M xi_src."""
    __slots__ = ("src", "m0", "M")

    def __init__(self, src, m0, M):
        self.src, self.m0, self.M = src, m0, M

    def left(self, L):
        return _Gen(self.src, L @ self.m0, None if self.M is None else L @ self.M)


class _Centre:
    """Affine centre: constant + one matrix per source."""
    __slots__ = ("m0", "parts")

    def __init__(self, m0, parts=None):
        self.m0, self.parts = m0, dict(parts or {})

    def left(self, L):
        return _Centre(L @ self.m0, {s: L @ M for s, M in self.parts.items()})

    def add(self, o):
        parts = dict(self.parts)
        for s, M in o.parts.items():
            parts[s] = parts[s] + M if s in parts else M
        return _Centre(self.m0 + o.m0, parts)

    def as_gens(self, L) -> List[_Gen]:
        """L @ centre as generator(s): single-source by construction (asserted)."""
        live = [(s, M) for s, M in self.parts.items() if np.any(M)]
        assert len(live) <= 1, "a generator would depend on two sources: not a TZDDPC tube chain"
        if not live:
            return [_Gen(SRC_NONE, L @ self.m0, None)]
        s, M = live[0]
        return [_Gen(s, L @ self.m0, L @ M)]


class _Zono:
    __slots__ = ("c", "gens")

    def __init__(self, c: _Centre, gens: List[_Gen]):
        self.c, self.gens = c, gens


def _mz_times(C, G, Z: _Zono) -> _Zono:
    """MatrixZonotope(C, G_i) * Z  =  [C Z, G_1 Z, ...]  (centre first, then the generators block by block)."""
    gens = [g.left(C) for g in Z.gens]
    for Gi in G:
        gens += Z.c.as_gens(Gi) + [g.left(Gi) for g in Z.gens]
    return _Zono(Z.c.left(C), gens)


def _plus(Z1: _Zono, Z2: _Zono) -> _Zono:
    return _Zono(Z1.c.add(Z2.c), Z1.gens + Z2.gens)


@dataclass
class GenStack:
    n: int
    m: int
    N: int
    nseg: int                  # tubes Ze[0] .. Ze[nseg - 1]
    seg_ptr: np.ndarray        # nseg + 1 offsets into the generator arrays (literal order inside a segment)
    src: np.ndarray            # G int32
    m0: np.ndarray             # G x n
    M: np.ndarray              # G x n x (n + m)   (columns beyond the width of the source are zero)
    c0: np.ndarray             # nseg x n                 centre: c0 + cE e0 + sum_j cZ[:, j] zeta_j
    cE: np.ndarray             # nseg x n x n
    cZ: np.ndarray             # nseg x N x n x (n + m)
    K: np.ndarray              # m x n

    @property
    def num_generators(self):
        return np.diff(self.seg_ptr)


def build_stack(MdataK, Mdelta, K, W, n: int, m: int, N: int, k0: Optional[int] = None, nseg: Optional[int] = None) -> GenStack:
    """Literal ``Ze[0 .. nseg-1]`` (default: the N tubes the constraints use) of ``build_problem`` (k0 None) or
    ``build_problem_simplified(k0)``, reference loops repeated line by line (including the ``range(1, k)`` nesting of ``:184``)."""
    p = n + m
    CK, GK = np.asarray(MdataK.center, float), np.asarray(MdataK.generators, float)
    Cd, Gd = np.asarray(Mdelta.center, float), np.asarray(Mdelta.generators, float)
    assert not np.any(Cd), "Mdelta must have a zero centre (reference tzddpc/tzddpc.py:122-123)"
    K = np.atleast_2d(np.asarray(K, float))
    zero = np.zeros(n)
    # Ze[0] = <e0, [0]>  (:172)                    XU[k] = <[xbar_k; v_k], [0]>  (:174)
    Ze0 = _Zono(_Centre(zero, {SRC_E0: np.eye(n)}), [_Gen(SRC_NONE, zero, None)])
    XU = [_Zono(_Centre(np.zeros(p), {1 + k: np.eye(p)}), [_Gen(SRC_NONE, np.zeros(p), None)]) for k in range(N)]
    Wz = _Zono(_Centre(np.asarray(W.center, float)), [_Gen(SRC_NONE, np.asarray(W.generators, float)[:, i], None) for i in range(W.generators.shape[1])])
    term1 = [_mz_times(CK, GK, Ze0)]                                           # :175
    Z_noise = [_plus(_mz_times(Cd, Gd, XU[k]), Wz) for k in range(N)]          # :176
    nseg = max(N, 2) if nseg is None else nseg          # Ze[1] is what solve() returns (:377), also for N = 1
    Ze = [Ze0]
    for k in range(nseg - 1):
        if k0 is None:
            term1.append(_mz_times(CK, GK, term1[-1]))                         # :181
            noise = Z_noise[0]                                                 # :183
            for j in range(1, k):                                              # :184
                noise = _plus(_mz_times(CK, GK, noise), Z_noise[j])            # :185
        else:
            term1.append(term1[-1] if k > k0 else _mz_times(CK, GK, term1[-1]))   # :292-295
            start = max(0, k - k0)                                             # :297
            noise = Z_noise[start]                                             # :298
            for j in range(1, min(k, k0)):                                     # :299
                noise = _plus(_mz_times(CK, GK, noise), Z_noise[start + j])    # :300
        Ze.append(_plus(term1[k], noise))                                      # :205-207 / :322-324
    G = sum(len(Z.gens) for Z in Ze)
    seg_ptr = np.zeros(nseg + 1, dtype=np.int64)
    src = np.full(G, SRC_NONE, dtype=np.int32); m0 = np.zeros((G, n)); M = np.zeros((G, n, p))
    c0 = np.zeros((nseg, n)); cE = np.zeros((nseg, n, n)); cZ = np.zeros((nseg, N, n, p))
    g = 0
    for k, Z in enumerate(Ze):
        c0[k] = Z.c.m0
        for s, Mc in Z.c.parts.items():
            if s == SRC_E0:
                cE[k] = Mc
            else:
                cZ[k, s - 1] = Mc
        for gen in Z.gens:
            src[g] = gen.src; m0[g] = gen.m0
            if gen.M is not None:
                M[g, :, :gen.M.shape[1]] = gen.M
            g += 1
        seg_ptr[k + 1] = g
    return GenStack(n, m, N, nseg, seg_ptr, src, m0, M, c0, cE, cZ, K)


def evaluate_host(st: GenStack, e0: np.ndarray, zeta: np.ndarray):
    """numpy evaluation of one trajectory (tests / documentation of what the kernel computes): (centre, rad_x, rad_u) per tube."""
    n, m = st.n, st.m
    xi = np.zeros((st.N + 2, n + m)); xi[1, :n] = e0; xi[2:] = zeta          # row 0: no source, 1: e0, 2 + j: zeta_j
    val = st.m0 + np.einsum("gic,gc->gi", st.M, xi[st.src + 1])
    out = []
    for k in range(st.nseg):
        sl = slice(st.seg_ptr[k], st.seg_ptr[k + 1])
        c = st.c0[k] + st.cE[k] @ e0 + np.einsum("jic,jc->i", st.cZ[k], zeta)
        out.append((c, np.abs(val[sl]).sum(axis=0), np.abs(val[sl] @ st.K.T).sum(axis=0)))
    return out


def restrict_to_e0(st: GenStack, nseg: int) -> GenStack:
    """The sub-stack of the generators that do not depend on a decision variable (constant or e0-sourced), tubes 0 .. nseg-1: what
    the device evaluates per solve for the literal problem (centres C_K^p e0 + c0 and the numeric part of the radii)."""
    keep, ptr = [], [0]
    for k in range(nseg):
        for g in range(int(st.seg_ptr[k]), int(st.seg_ptr[k + 1])):
            if st.src[g] <= 0:
                keep.append(g)
        ptr.append(len(keep))
    keep = np.asarray(keep, dtype=np.int64)
    return GenStack(st.n, st.m, st.N, nseg, np.asarray(ptr, dtype=np.int64), st.src[keep], st.m0[keep], st.M[keep],
                    st.c0[:nseg], st.cE[:nseg], np.zeros_like(st.cZ[:nseg]), st.K)


def count_generators(gK: int, gD: int, gW: int, N: int, k0: Optional[int] = None, nseg: Optional[int] = None):
    """Generator counts of Ze[0 .. nseg-1] and how many of them depend on a decision variable, from the recurrences of
    ``build_stack`` alone (nothing is materialised): (total per tube, decision-dependent per tube)."""
    nseg = max(N, 2) if nseg is None else nseg
    t1 = [(gK + 1) * 2 - 1]                          # MdataK * <e0, [0]>: (gamma + 1)(g + 1) - 1 with g = 1
    zn = gD + 1 + gW                                 # Mdelta * <zeta, [0]> + W : (gD + 1) * 2 - 1 + gW ... the centre column is zero but kept
    zn = (gD + 1) * 2 - 1 + gW
    tot, dec = [1], [0]
    for k in range(nseg - 1):
        if k0 is None:
            t1.append((gK + 1) * (t1[-1] + 1) - 1)
            noise = zn
            for _ in range(1, k):
                noise = (gK + 1) * (noise + 1) - 1 + zn
        else:
            t1.append(t1[-1] if k > k0 else (gK + 1) * (t1[-1] + 1) - 1)
            noise = zn
            for _ in range(1, min(k, k0)):
                noise = (gK + 1) * (noise + 1) - 1 + zn
        tot.append(t1[k] + noise)
        dec.append(noise)                            # upper bound: the W columns inside are constants
    return tot, dec
