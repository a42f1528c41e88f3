"""Build-time assembly of the TZDDPC problem into the data the HIP kernels consume.

Replaces the symbolic construction in reference ``tzddpc/tzddpc.py:132-241`` (``build_problem``)
and ``:243-355`` (``build_problem_simplified``).  Runs once per ``build_problem`` call on the host
(numpy); nothing here is on the per-step path.

What the reference does literally -- ``MatrixZonotope * CVXZonotope`` products whose generator
count grows by (gamma_K + 1) per horizon step -- is collapsed exactly:
when every generator of ``MdataK`` / ``Mdelta`` has a single non-zero entry (the Girard order-1
boxes of ``:126-128``), with Delta = sum_i |G_i| (entry-wise),

    M_K * <c, D, [(P_l, b_l)]> = < C_K c, C_K D, [(C_K P_l, b_l)] + [(I, Delta_K(|c| + rad))] >
    rad = sum_j |D_:j| + sum_l |P_l| b_l                         (interval radius, ``:191``)

so every interval radius is  (a part depending on e0 only, evaluated per step by the HIP kernel
``tz_prepare``)  +  (a constant from W)  +  (non-negative matrices) x |[xbar_j; v_j]|.  The only
decision-dependent non-linearity is the component-wise absolute value, handled by epigraph
variables t_j >= |[xbar_j; v_j]|.  The result is a QP whose matrices are shared by all
trajectories and all MPC steps; only q, l, u depend -- affinely -- on

    theta = [ xbar0 | |xbar0| | (c_k, rho^x_k, rho^u_k)_{k<N} ]      (the last block from e0)

Decision vector (condensed, xbar eliminated through xbar_k = A^k xbar0 + sum_j A^(k-1-j) B v_j):

    z = [ v (N m) | t (used |.| epigraphs) | s (loss epigraphs) | u_free (only if the loss needs it) ]
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

from . import cplite
from .cplite import Affine, Constraint, Convex, CpliteError


class StructureError(NotImplementedError):
    """Matrix-zonotope generators are not single-entry: needs the general stacked-generator path."""


@dataclass
class TubeConstants:
    n: int
    m: int
    N: int
    CK: np.ndarray            # center of MdataK  (= Ahat + Bhat K)
    DK: np.ndarray            # Delta_K   n x n
    K: np.ndarray             # m x n
    pmax: int                 # highest power of M_K applied to <e0>
    absCKpow: np.ndarray      # (pmax, n, n)   |C_K^j|
    absKCKpow: np.ndarray     # (pmax, m, n)   |K C_K^j|
    power: np.ndarray         # (N,) int32: Ze_k's e0-part is M_K^power[k] <e0>


@dataclass
class ParametricQP:
    n: int
    m: int
    N: int
    nz: int
    nc: int
    ntheta: int
    P: np.ndarray
    A: np.ndarray
    # affine maps over theta (dense here; the native layer converts to CSR)
    q0: np.ndarray
    Qt: np.ndarray            # nz x ntheta
    l0: np.ndarray
    Lt: np.ndarray            # nc x ntheta
    u0: np.ndarray
    Ut: np.ndarray
    # pure-parameter rows  pl <= f0 + Ft theta <= pu
    f0: np.ndarray
    Ft: np.ndarray
    pl: np.ndarray
    pu: np.ndarray
    # objective constant r0 + r1'xbar0 + xbar0'R2 xbar0
    r0: float
    r1: np.ndarray
    R2: np.ndarray
    # recovery  xbar = Phi xbar0 + Gam v
    Phi: np.ndarray
    Gam: np.ndarray
    tube: TubeConstants
    row_names: List[str] = field(default_factory=list)
    var_names: List[str] = field(default_factory=list)
    # constants for the literal Ze[1] export (reference returns Ze[1], tzddpc.py:377)
    n_v: int = 0
    # part of the objective constant that is linear in theta beyond xbar0 (solve_simplified2's regulariser only; added on the host)
    rt: Optional[np.ndarray] = None

    def theta_index(self):
        n, m, N = self.n, self.m, self.N
        blk = 2 * n + m
        return dict(xbar0=lambda i: i, absx0=lambda i: n + i,
                    c=lambda k, i: 2 * n + k * blk + i,
                    rx=lambda k, i: 2 * n + k * blk + n + i,
                    ru=lambda k, j: 2 * n + k * blk + 2 * n + j)


class _Box:
    """Axis-aligned box half-widths  b = b0 + Bt t  (t = stacked |[xbar_j; v_j]|, j < N)."""
    __slots__ = ("b0", "Bt")

    def __init__(self, b0, Bt):
        self.b0, self.Bt = b0, Bt


class _Z:
    """< c, D, [(P_l, box_l)] > with numeric c, D, P_l."""
    __slots__ = ("c", "D", "layers")

    def __init__(self, c, D, layers):
        self.c, self.D, self.layers = c, D, layers

    def radius(self, L, nt):
        b0 = np.abs(L @ self.D).sum(axis=1)
        Bt = np.zeros((L.shape[0], nt))
        for P, box in self.layers:
            M = np.abs(L @ P)
            b0 = b0 + M @ box.b0
            Bt = Bt + M @ box.Bt
        return b0, Bt


def _apply_MK(CK, DK, Z: _Z, nt) -> _Z:
    n = CK.shape[0]
    b0, Bt = Z.radius(np.eye(n), nt)
    new = _Box(DK @ (np.abs(Z.c) + b0), DK @ Bt)
    return _Z(CK @ Z.c, CK @ Z.D, [(CK @ P, box) for P, box in Z.layers] + [(np.eye(n), new)])


def _minkowski(Z1: _Z, Z2: _Z) -> _Z:
    return _Z(Z1.c + Z2.c, np.concatenate([Z1.D, Z2.D], axis=1), Z1.layers + Z2.layers)


def noise_schedule(N: int, k0: Optional[int]):
    """For k' = 0..N-1: (start, J) with term2[k'] = sum_{i<=J} M_K^(J-i) Z_noise[start+i]  (literal nesting).

    full (``:183-186``): start 0, J = max(k'-1, 0);  simplified (``:297-300``): start = max(0,k'-k0),
    J = max(min(k',k0)-1, 0).
    """
    out = []
    for kp in range(N):
        if k0 is None:
            out.append((0, max(kp - 1, 0)))
        else:
            out.append((max(0, kp - k0), max(min(kp, k0) - 1, 0)))
    return out


def term1_power(N: int, k0: Optional[int]) -> np.ndarray:
    """power[k]: Ze_k's e0 part is M_K^power[k] <e0>; term1[j] = M_K^(min(j, k0+1)+1) <e0> (``:175,181,292-295``)."""
    pw = np.zeros(N, dtype=np.int32)
    for k in range(1, N):
        j = k - 1
        pw[k] = (j + 1) if k0 is None else (min(j, k0 + 1) + 1)
    return pw


def rank_one_factor(Dd: np.ndarray, tol: float = 1e-12):
    """If Dd = rho s' (what boxing a matrix zonotope built from W x pinv(data) gives: Dd[r,c] = rad(W)_r sum_t|P[t,c]|) return s
    normalised to max(s) = 1 and the pivot column, else None.  Exact up to rounding: relative residual <= tol."""
    if not np.any(Dd):
        return None
    r0, c0 = np.unravel_index(np.argmax(np.abs(Dd)), Dd.shape)
    s = Dd[r0] / Dd[r0, c0]
    rho = Dd[:, c0]
    if np.abs(Dd - np.outer(rho, s)).max() <= tol * np.abs(Dd).max() and np.all(s >= 0):
        return s, int(c0)
    return None


def newton_cost(qp: "ParametricQP") -> float:
    """Cost of one Newton system of the interior point for this formulation: sparse outer products of G'WG, Cholesky, and a
    per-row term (element-wise work, two slots of LDS and a pass of the ELL products per row) -- used to choose between equivalent
    formulations; only the order matters."""
    fu = np.isfinite(qp.u0); fl = np.isfinite(qp.l0)
    nnz = np.count_nonzero(qp.A, axis=1)
    rows = np.concatenate([nnz[fu], nnz[fl]])
    return float((rows.astype(float) ** 2).sum() + qp.nz ** 3 / 3.0 + 100.0 * rows.size)


def build_parametric_qp(Ahat, Bhat, CK, DK, Dd, K, W_c, W_G, xl, xu, ul, uu, N: int,
                        build_loss: Callable, build_constraints: Optional[Callable],
                        k0: Optional[int] = None, epigraph: str = "auto", literal=None, cuts=None) -> ParametricQP:
    """epigraph: "component" -- one variable t_{j,c} >= |zeta_{j,c}| per used component (2 rows each);
                 "aggregate" -- when Delta^delta is rank one (rho s'), every radius depends on t_j only through
                                tau_j = s'|zeta_j|: one variable per step j and 2^(#components) sign rows
                                tau_j >= sum_c (+-) s_c zeta_{j,c}.  Same feasible (v, xbar) set, far fewer variables;
                 "group:g"   -- the same with the components dealt into groups of at most g: tau_j = sum_G tau_{j,G},
                                tau_{j,G} >= sum_{c in G} (+-) s_c zeta_{j,c}: ceil(p / g) variables and at most 2^g rows per
                                group and step (g = p is "aggregate", g = 1 the component form written in tau);
                 "auto"      -- rank one: the group size with the cheapest Newton system (`newton_cost` of the assembled
                                candidates), else "component".
    literal: a `genstack.GenStack` of the tubes Ze[0..N-1] -- the LITERAL problem for matrix zonotopes with dense generators
             (no collapse possible): one epigraph variable per entry of every generator column that depends on a decision
             variable, `t >= |m0_i + M_i zeta_j|` (what cvxpy's canonicalisation of ``.interval`` does, reference ``:191-197``);
             the generators that depend on e0 only stay numeric: their radii reach the problem through theta's rho entries,
             which the device then takes from the stack (``tz_problem_attach_tube_stack``), not from the collapsed recursion.
             DK / Dd are ignored.  Sized by the caller: N * (generators per tube) * (n + m) variables.
    cuts:    (with `literal`) the CUTTING-PLANE form of the literal problem: no epigraph variables at all; the radius of a tube row,
             sum_g |a_g(v, xbar0)| over its decision-dependent generators, is replaced by sum_g sigma_g a_g for every sign pattern
             sigma in cuts[(k, "x" | "u", i)] -- each one a valid inequality for every parameter value (sum sigma a <= sum |a|), so
             the rows are shared by all trajectories and steps like the rest of G; the caller adds the patterns that the literal
             evaluation of a solution shows to be violated until none is (`TZDDPC` does, with K1g as the separation routine).  The
             problem keeps N m + (loss epigraphs) variables whatever the generator count.  `ParametricQP.families` lists, per
             (k, kind, i), the generators of the row (indices into the tube's literal order) for the caller's sign look-up."""
    Ahat = np.asarray(Ahat, float); Bhat = np.asarray(Bhat, float)
    K = np.atleast_2d(np.asarray(K, float))
    n, m = Bhat.shape
    p = n + m
    nt = N * p
    W_c = np.asarray(W_c, float).reshape(n); W_G = np.asarray(W_G, float).reshape(n, -1)

    # ---- condensed nominal dynamics (``:166-170``): xbar_k = Phi_k xbar0 + Gam_k v ------------
    Phi = np.zeros((N + 1, n, n)); Gam = np.zeros((N + 1, n, N * m))
    Phi[0] = np.eye(n)
    for k in range(N):
        Phi[k + 1] = Ahat @ Phi[k]
        Gam[k + 1] = Ahat @ Gam[k]
        Gam[k + 1][:, k * m:(k + 1) * m] += Bhat

    if literal is not None:
        DK = np.zeros((n, n)); Dd = np.zeros((n, p)); epigraph = "component"
    # ---- noise chains term2[k'] (e0-independent; numeric centers, boxes affine in t) -----------
    Z_noise = []
    for j in range(N):                                                   # ``:176``
        Bt = np.zeros((n, nt)); Bt[:, j * p:(j + 1) * p] = Dd
        Z_noise.append(_Z(W_c.copy(), W_G.copy(), [(np.eye(n), _Box(np.zeros(n), Bt))]))
    term2 = []
    for start, J in noise_schedule(N, k0):
        Zn = Z_noise[start]
        for i in range(1, J + 1):
            Zn = _minkowski(_apply_MK(CK, DK, Zn, nt), Z_noise[start + i])
        term2.append(Zn)
    power = term1_power(N, k0)
    pmax = int(power.max(initial=0))
    absCK = np.zeros((max(pmax, 1), n, n)); absKCK = np.zeros((max(pmax, 1), m, n))
    Mp = np.eye(n)
    for j in range(max(pmax, 1)):
        absCK[j] = np.abs(Mp); absKCK[j] = np.abs(K @ Mp)
        Mp = CK @ Mp
    tube = TubeConstants(n, m, N, CK.copy(), DK.copy(), K.copy(), pmax, absCK, absKCK, power)

    # per-step noise-side center / radii
    cn = np.zeros((N, n)); rx0 = np.zeros((N, n)); rxT = np.zeros((N, n, nt)); ru0 = np.zeros((N, m)); ruT = np.zeros((N, m, nt))
    for k in range(1, N):
        Zn = term2[k - 1]
        cn[k] = Zn.c
        rx0[k], rxT[k] = Zn.radius(np.eye(n), nt)
        ru0[k], ruT[k] = Zn.radius(K, nt)

    lit_x, lit_u = [], []             # literal mode: (tube k, row i, generator g) of every decision-dependent entry
    if literal is not None:
        st = literal
        assert st.nseg >= N and st.n == n and st.m == m
        # the constant part c0 of the tube centres (W.center propagated) reaches the rows through theta's centre slot, which the
        # device / theta_reference fill with c0 + C_K^p e0 from the e0-restricted stack: it must not be subtracted here as well
        cn[:] = 0.0; rx0[:] = 0.0; ru0[:] = 0.0; rxT[:] = 0.0; ruT[:] = 0.0
        Mp_ = np.eye(n)
        pw_mats = [Mp_]
        for _ in range(pmax):
            pw_mats.append(CK @ pw_mats[-1])
        for k in range(N):
            assert np.allclose(st.cE[k], pw_mats[int(power[k])], atol=1e-12), "centre chain of the stack is not C_K^power"
            assert not np.any(st.cZ[k]), "tube centres that depend on the decision variables are not expected (Mdelta has a zero centre)"
            for g in range(int(st.seg_ptr[k]), int(st.seg_ptr[k + 1])):
                if st.src[g] <= 0:
                    continue                                  # constant or e0-sourced: numeric per solve (theta)
                for i in range(n):
                    if np.any(st.M[g, i]):
                        lit_x.append((k, i, g))
                KM = K @ st.M[g]
                for j2 in range(m):
                    if np.any(KM[j2]):
                        lit_u.append((k, j2, g))
    families = {}
    if cuts is not None:
        assert literal is not None, "cuts need the literal stack"
        for (k, i, g) in lit_x:
            families.setdefault((k, "x", i), []).append(g)
        for (k, j2, g) in lit_u:
            families.setdefault((k, "u", j2), []).append(g)
        lit_x, lit_u = [], []                                # no epigraph variables: sign-pattern rows instead
    # ---- which |.| epigraphs are needed --------------------------------------------------------
    used = (np.abs(rxT).sum(axis=(0, 1)) + np.abs(ruT).sum(axis=(0, 1))) > 0      # (nt,)
    t_var = {}                       # (j, c) -> z index            (component form)
    tau_var = {}                     # j -> z index                  (aggregate form)
    var_names = [f"v[{k},{j}]" for k in range(N) for j in range(m)]
    nzc = N * m
    rk1 = rank_one_factor(Dd) if epigraph != "component" else None
    if epigraph != "auto" and epigraph != "component" and rk1 is None:
        raise StructureError("aggregate epigraphs need a rank-one Delta^delta (boxed W x data structure)")
    aggregate = rk1 is not None
    gsize = 0
    if aggregate:
        s_w, c_piv = rk1
        ncomp = int(np.count_nonzero(s_w > 0))
        if epigraph == "aggregate":
            gsize = ncomp
        elif epigraph.startswith("group:"):
            gsize = max(1, min(int(epigraph.split(":", 1)[1]), ncomp))
        else:
            # "auto": the equivalent formulations are assembled and the cheapest Newton system wins (build time only)
            best = None
            for g in range(1, ncomp + 1):
                if 2 ** g > 64:
                    break
                cand = build_parametric_qp(Ahat, Bhat, CK, DK, Dd, K, W_c, W_G, xl, xu, ul, uu, N, build_loss, build_constraints, k0, f"group:{g}")
                c = newton_cost(cand)
                if best is None or c < best[0]:
                    best = (c, cand)
            return best[1]
        for j in range(N):
            if np.any(used[j * p:(j + 1) * p]):
                comps = [c for c in range(p) if s_w[c] > 0 and not (j == 0 and c < n)]   # |xbar0| is a parameter
                groups = [comps[i:i + gsize] for i in range(0, len(comps), gsize)] or [[]]
                tau_var[j] = []
                for gi, grp in enumerate(groups):
                    tau_var[j].append((nzc, grp)); nzc += 1
                    var_names.append(f"tau[{j}]" if len(groups) == 1 else f"tau[{j},{gi}]")
    else:
        for j in range(N):
            for c in range(p):
                if used[j * p + c] and not (j == 0 and c < n):       # |xbar0| is a parameter
                    t_var[(j, c)] = nzc; nzc += 1
                    var_names.append(f"t[{j},{c}]")

    lit_x0 = nzc
    nzc += len(lit_x)
    var_names += [f"lx[{k},{i},{g}]" for k, i, g in lit_x]
    lit_u0 = nzc
    nzc += len(lit_u)
    var_names += [f"lu[{k},{j},{g}]" for k, j, g in lit_u]
    # ---- callbacks on look-alike variables -----------------------------------------------------
    nsym = N * m + N * m + n                       # [v | u_free | xbar0]
    sv = slice(0, N * m); su = slice(N * m, 2 * N * m); sp = slice(2 * N * m, nsym)
    Cv = np.zeros((N * m, nsym)); Cv[:, sv] = np.eye(N * m)
    v_expr = Affine(Cv, np.zeros(N * m), (N, m))
    Cu = np.zeros((N * m, nsym)); Cu[:, su] = np.eye(N * m)
    u_expr = Affine(Cu, np.zeros(N * m), (N, m))
    Cx = np.zeros(((N + 1) * n, nsym))
    Cx[:, sv] = Gam.reshape((N + 1) * n, N * m); Cx[:, sp] = Phi.reshape((N + 1) * n, n)
    xbar_expr = Affine(Cx, np.zeros((N + 1) * n), (N + 1, n))
    if k0 is None:
        loss = build_loss(u_expr, xbar_expr)                                        # ``:222``
        cons = build_constraints(v_expr, xbar_expr) if build_constraints is not None else []   # ``:213``
    else:
        loss = build_loss(v_expr, xbar_expr[1:])                                    # ``:336``
        cons = build_constraints(v_expr, xbar_expr[1:]) if build_constraints is not None else []  # ``:327``
    if loss is None:
        raise Exception("Loss function is not defined or is not convex!")
    loss = cplite.as_convex(loss, nsym)
    cons = [] if cons is None else list(cons)
    for idx, c in enumerate(cons):
        if c is None or not isinstance(c, Constraint):
            raise Exception(f"Constraint {idx} is not defined or is not convex.")           # ``:215-217``

    # free-u analysis (``:160,222``: u is constrained by nothing)
    def _uses_u(C):
        return bool(np.any(C[:, su]))

    def _only_u_homogeneous(e: Affine):
        return _uses_u(e.C) and not np.any(e.C[:, sv]) and not np.any(e.C[:, sp]) and not np.any(e.d)

    if np.any(loss.lin[su]):
        raise Exception("Problem is unbounded")     # a linear term in the free variable u
    need_u = any(_uses_u(c.expr.C) for c in cons)
    terms = {"sq": [], "ab": [], "mx": []}
    for kind in ("sq", "ab", "mx"):
        for w, e in getattr(loss, kind):
            if w == 0.0:
                continue
            if _only_u_homogeneous(e):
                continue                             # min over free u of a norm of a linear map of u is 0
            if _uses_u(e.C):
                need_u = True
            terms[kind].append((w, e))
    for w, e in loss.soc:
        # ||F u||_2 of the FREE variable u (reference :160, :222: no constraint mentions it) is minimised to 0 independently of
        # everything else -- exact as long as no other term ties u down; a cone on anything else has no QP form
        if w < 0.0:
            raise CpliteError("a second-order-cone term with a negative weight is not convex (the reference's is_dcp() check, "
                              "tzddpc/tzddpc.py:224, rejects it)")
        if w != 0.0 and (need_u or not _only_u_homogeneous(e)):
            raise CpliteError(cplite.SOC_MESSAGE)
    u_var0 = None
    if need_u:
        u_var0 = nzc; nzc += N * m
        var_names += [f"u[{k},{j}]" for k in range(N) for j in range(m)]

    def z_coef(Crow, nz_now):
        """symbol coefficients -> (z-part over current nz, xbar0-part)."""
        zc = np.zeros(nz_now)
        zc[:N * m] = Crow[sv]
        if u_var0 is not None:
            zc[u_var0:u_var0 + N * m] = Crow[su]
        return zc, Crow[sp]

    # loss epigraph variables
    epi_specs = []       # (weight, [(Crow, d)], "sum"|"max")
    for w, e in terms["ab"]:
        for r in range(e.size):
            if not np.any(e.C[r]):
                continue
            epi_specs.append((w, [(e.C[r], e.d[r])]))
    for w, e in terms["mx"]:
        rows_ = [(e.C[r], e.d[r]) for r in range(e.size)]
        if any(np.any(c) for c, _ in rows_):
            epi_specs.append((w, rows_))
    s_var0 = nzc
    nzc += len(epi_specs)
    var_names += [f"s[{i}]" for i in range(len(epi_specs))]
    nz = nzc

    ntheta = 2 * n + N * (2 * n + m)
    blk = 2 * n + m
    ix_x0 = lambda i: i
    ix_ax0 = lambda i: n + i
    ix_c = lambda k, i: 2 * n + k * blk + i
    ix_rx = lambda k, i: 2 * n + k * blk + n + i
    ix_ru = lambda k, j: 2 * n + k * blk + 2 * n + j

    rows = []   # (name, zcoef, lo_c, lo_t, hi_c, hi_t)

    def add_row(name, zc, lo_c, lo_t, hi_c, hi_t):
        rows.append((name, zc, lo_c, lo_t, hi_c, hi_t))

    zeros_t = lambda: np.zeros(ntheta)

    def gen_row(g, i, left):
        """a_g = c + f.z + th.theta for row i of generator g (left = None) or of K g (left = K): what the literal epigraphs bound"""
        j = int(literal.src[g]) - 1
        rowM = literal.M[g][i] if left is None else left[i] @ literal.M[g]
        c_ = float(literal.m0[g][i] if left is None else left[i] @ literal.m0[g])
        f = np.zeros(nz); f[:N * m] = rowM[:n] @ Gam[j]; f[j * m:(j + 1) * m] += rowM[n:]
        th = zeros_t(); th[[ix_x0(cc) for cc in range(n)]] = rowM[:n] @ Phi[j]
        return c_, f, th

    def cut_terms(key, sigma):
        """sum_g sigma_g a_g over the family `key`: (constant, z coefficients, theta coefficients)"""
        k, kind, i = key
        c_s, f_s, th_s = 0.0, np.zeros(nz), zeros_t()
        for sg, g in zip(sigma, families[key]):
            c_, f, th = gen_row(g, i, None if kind == "x" else K)
            c_s += sg * c_; f_s += sg * f; th_s += sg * th
        return c_s, f_s, th_s

    # ---- tube rows (``:189-209``) --------------------------------------------------------------
    for k in range(N):
        for i in range(n):
            zc = np.zeros(nz); zc[:N * m] = Gam[k][i]
            tt = np.zeros(nz)
            th_abs = zeros_t()
            if aggregate:
                for j, grps in tau_var.items():
                    for zi, _ in grps:
                        tt[zi] = rxT[k][i, j * p + c_piv]        # radius coefficient on tau_j = sum_G tau_{j,G} (s normalised to s[c_piv] = 1)
            else:
                for (j, c), zi in t_var.items():
                    tt[zi] = rxT[k][i, j * p + c]
                for c in range(n):
                    th_abs[ix_ax0(c)] = rxT[k][i, c]             # j = 0, xbar part -> |xbar0|
            for q_, (k2, i2, _) in enumerate(lit_x):
                if k2 == k and i2 == i:
                    tt[lit_x0 + q_] = 1.0
            base = zeros_t()
            base[[ix_x0(c) for c in range(n)]] = Phi[k][i]
            base[ix_c(k, i)] += 1.0
            rxk = zeros_t(); rxk[ix_rx(k, i)] = 1.0
            # upper: xbar + c + rad <= xu
            add_row(f"Xub[{k},{i}]", zc + tt, -np.inf, zeros_t(), xu[i] - cn[k, i] - rx0[k, i], -(base + rxk + th_abs))
            # lower: xbar + c - rad >= xl
            add_row(f"Xlb[{k},{i}]", zc - tt, xl[i] - cn[k, i] + rx0[k, i], -(base - rxk - th_abs), np.inf, zeros_t())
            for ci_, sigma in enumerate((cuts or {}).get((k, "x", i), [])):     # cutting planes: rad >= sum_g sigma_g a_g
                c_s, f_s, th_s = cut_terms((k, "x", i), sigma)
                add_row(f"Xub[{k},{i}]c{ci_}", zc + f_s, -np.inf, zeros_t(), xu[i] - cn[k, i] - rx0[k, i] - c_s, -(base + rxk + th_abs + th_s))
                add_row(f"Xlb[{k},{i}]c{ci_}", zc - f_s, xl[i] - cn[k, i] + rx0[k, i] + c_s, -(base - rxk - th_abs - th_s), np.inf, zeros_t())
        Kcn = K @ cn[k]
        for j2 in range(m):
            zc = np.zeros(nz); zc[k * m + j2] = 1.0
            tt = np.zeros(nz); th_abs = zeros_t()
            if aggregate:
                for j, grps in tau_var.items():
                    for zi, _ in grps:
                        tt[zi] = ruT[k][j2, j * p + c_piv]
            else:
                for (j, c), zi in t_var.items():
                    tt[zi] = ruT[k][j2, j * p + c]
                for c in range(n):
                    th_abs[ix_ax0(c)] = ruT[k][j2, c]
            for q_, (k2, j3, _) in enumerate(lit_u):
                if k2 == k and j3 == j2:
                    tt[lit_u0 + q_] = 1.0
            base = zeros_t()
            for i in range(n):
                base[ix_c(k, i)] += K[j2, i]
            ruk = zeros_t(); ruk[ix_ru(k, j2)] = 1.0
            add_row(f"Uub[{k},{j2}]", zc + tt, -np.inf, zeros_t(), uu[j2] - Kcn[j2] - ru0[k, j2], -(base + ruk + th_abs))
            add_row(f"Ulb[{k},{j2}]", zc - tt, ul[j2] - Kcn[j2] + ru0[k, j2], -(base - ruk - th_abs), np.inf, zeros_t())
            for ci_, sigma in enumerate((cuts or {}).get((k, "u", j2), [])):
                c_s, f_s, th_s = cut_terms((k, "u", j2), sigma)
                add_row(f"Uub[{k},{j2}]c{ci_}", zc + f_s, -np.inf, zeros_t(), uu[j2] - Kcn[j2] - ru0[k, j2] - c_s, -(base + ruk + th_abs + th_s))
                add_row(f"Ulb[{k},{j2}]c{ci_}", zc - f_s, ul[j2] - Kcn[j2] + ru0[k, j2] + c_s, -(base - ruk - th_abs - th_s), np.inf, zeros_t())
    # ---- tau_j >= s'|zeta_j|  (aggregate form): one row per sign pattern of the components with s_c > 0 -------------
    if aggregate:
        import itertools
        for j, grps in tau_var.items():
            par_c = [c for c in range(p) if s_w[c] > 0 and (j == 0 and c < n)]   # xbar0 components: |xbar0| is a parameter
            for gi, (zi, var_c) in enumerate(grps):                            # var_c: components that depend on decision variables
                vname = var_names[zi]
                for signs in itertools.product((1.0, -1.0), repeat=len(var_c)):
                    zrow = np.zeros(nz); zrow[zi] = 1.0
                    th = zeros_t()
                    for sg, c in zip(signs, var_c):
                        if c < n:
                            zrow[:N * m] -= sg * s_w[c] * Gam[j][c]
                            th[[ix_x0(cc) for cc in range(n)]] += sg * s_w[c] * Phi[j][c]
                        else:
                            zrow[j * m + (c - n)] -= sg * s_w[c]
                    if gi == 0:
                        for c in par_c:
                            th[ix_ax0(c)] += s_w[c]
                    add_row(f"{vname}{''.join('+' if sg > 0 else '-' for sg in signs)}", zrow, 0.0, th, np.inf, zeros_t())
    # ---- literal epigraphs: t >= |L (m0 + M zeta_j)| row by row, zeta_j = [Phi_j xbar0 + Gam_j v ; v_j] -------------------------
    for base_, entries, left in ((lit_x0, lit_x, None), (lit_u0, lit_u, K)):
        for q_, (k, i, g) in enumerate(entries):
            j = int(literal.src[g]) - 1
            rowM = literal.M[g][i] if left is None else left[i] @ literal.M[g]
            c_ = float(literal.m0[g][i] if left is None else left[i] @ literal.m0[g])
            f = np.zeros(nz); f[:N * m] = rowM[:n] @ Gam[j]; f[j * m:(j + 1) * m] += rowM[n:]
            th = zeros_t(); th[[ix_x0(cc) for cc in range(n)]] = rowM[:n] @ Phi[j]
            e_t = np.zeros(nz); e_t[base_ + q_] = 1.0
            nm = var_names[base_ + q_]
            add_row(f"{nm}+", e_t - f, c_, th, np.inf, zeros_t())           # t - f(v) >= c + th.theta
            add_row(f"{nm}-", e_t + f, -c_, -th, np.inf, zeros_t())
    # ---- t >= |zeta| (component form) ----------------------------------------------------------
    for (j, c), zi in t_var.items():
        zeta = np.zeros(nz); zt = zeros_t()
        if c < n:
            zeta[:N * m] = Gam[j][c]
            zt[[ix_x0(cc) for cc in range(n)]] = Phi[j][c]
        else:
            zeta[j * m + (c - n)] = 1.0
        e_t = np.zeros(nz); e_t[zi] = 1.0
        add_row(f"t+[{j},{c}]", e_t - zeta, 0.0, zt, np.inf, zeros_t())      # t - zeta >= +theta part
        add_row(f"t-[{j},{c}]", e_t + zeta, 0.0, -zt, np.inf, zeros_t())
    # ---- loss ----------------------------------------------------------------------------------
    P = np.zeros((nz, nz)); q0 = np.zeros(nz); Qt = np.zeros((nz, ntheta))
    r0 = float(loss.const); r1 = loss.lin[sp].copy(); R2 = np.zeros((n, n))
    zc, _ = z_coef(loss.lin, nz); q0 += zc
    for w, e in terms["sq"]:
        F = np.zeros((e.size, nz)); G = np.zeros((e.size, n))
        for r in range(e.size):
            F[r], G[r] = z_coef(e.C[r], nz)
        P += 2.0 * w * F.T @ F
        q0 += 2.0 * w * F.T @ e.d
        Qt[:, :n] += 2.0 * w * F.T @ G
        r0 += w * float(e.d @ e.d); r1 += 2.0 * w * G.T @ e.d; R2 += w * G.T @ G
    for w, e in terms["ab"]:
        for r in range(e.size):
            if not np.any(e.C[r]):
                r0 += w * abs(float(e.d[r]))
    for si, (w, rows_) in enumerate(epi_specs):
        zi = s_var0 + si
        q0[zi] += w
        for Crow, d in rows_:
            zc, pc = z_coef(Crow, nz)
            e_s = np.zeros(nz); e_s[zi] = 1.0
            th = zeros_t(); th[:n] = pc
            add_row(f"s+[{si}]", e_s - zc, d, th, np.inf, zeros_t())
            add_row(f"s-[{si}]", e_s + zc, -d, -th, np.inf, zeros_t())
    # ---- user constraints ----------------------------------------------------------------------
    for ci, c in enumerate(cons):
        e = c.expr
        for r in range(e.size):
            zc, pc = z_coef(e.C[r], nz)
            th = zeros_t(); th[:n] = -pc
            d = -float(e.d[r])
            if c.kind == "<=":
                add_row(f"user[{ci},{r}]", zc, -np.inf, zeros_t(), d, th)
            elif c.kind == ">=":
                add_row(f"user[{ci},{r}]", zc, d, th, np.inf, zeros_t())
            else:
                add_row(f"user[{ci},{r}]", zc, d, th, d, th.copy())

    qp_rows = [r for r in rows if np.any(r[1])]
    pr_rows = [r for r in rows if not np.any(r[1])]
    nc = len(qp_rows)
    A = np.array([r[1] for r in qp_rows]).reshape(nc, nz)
    l0 = np.array([r[2] for r in qp_rows]); Lt = np.array([r[3] for r in qp_rows]).reshape(nc, ntheta)
    u0 = np.array([r[4] for r in qp_rows]); Ut = np.array([r[5] for r in qp_rows]).reshape(nc, ntheta)
    # parameter-only rows  0 in [lo, hi]  ->  pl <= Ft theta <= pu  with  value := -(theta part)
    npr = len(pr_rows)
    Ft = np.zeros((2 * npr, ntheta)); f0 = np.zeros(2 * npr); pl = np.full(2 * npr, -np.inf); pu = np.full(2 * npr, np.inf)
    for i, (name, _, lo_c, lo_t, hi_c, hi_t) in enumerate(pr_rows):
        # need lo_c + lo_t.theta <= 0 <= hi_c + hi_t.theta
        f0[2 * i] = lo_c if np.isfinite(lo_c) else 0.0; Ft[2 * i] = lo_t; pu[2 * i] = 0.0 if np.isfinite(lo_c) else np.inf
        f0[2 * i + 1] = hi_c if np.isfinite(hi_c) else 0.0; Ft[2 * i + 1] = hi_t; pl[2 * i + 1] = 0.0 if np.isfinite(hi_c) else -np.inf
    keep = np.isfinite(pl) | np.isfinite(pu)
    Ft, f0, pl, pu = Ft[keep], f0[keep], pl[keep], pu[keep]

    out = ParametricQP(n=n, m=m, N=N, nz=nz, nc=nc, ntheta=ntheta, P=0.5 * (P + P.T), A=A,
                       q0=q0, Qt=Qt, l0=l0, Lt=Lt, u0=u0, Ut=Ut, f0=f0, Ft=Ft, pl=pl, pu=pu,
                       r0=r0, r1=r1, R2=R2, Phi=Phi.reshape((N + 1) * n, n), Gam=Gam.reshape((N + 1) * n, N * m),
                       tube=tube, row_names=[r[0] for r in qp_rows], var_names=var_names, n_v=N * m)
    if literal is not None:
        from .genstack import restrict_to_e0
        out.estack = restrict_to_e0(literal, N)
    if cuts is not None:
        out.families = families
    return out


def build_simplified2_qp(Acl, Bhat, K, deltaA, deltaB, W_c, W_G, Zsigma, Xz, Uz, N: int,
                         build_loss: Callable, build_constraints: Optional[Callable], ze_sum: str = "radius") -> ParametricQP:
    """The problem of ``TZDDPC.solve_simplified2`` (reference ``tzddpc/tzddpc.py:381-500``) as a two-sided parametric QP in
    x = [v | beta_u | beta_x | loss epigraphs], theta as in `build_parametric_qp` (only xbar0 and the tube centres C_K^k e0 are
    used).  xbar and ubar are condensed (``:424-428, :445``: xbar_{k+1} = Acl xbar_k + B v_k, ubar_k = K xbar_k + v_k); the
    zonotope memberships of ``:421-422`` stay as equality rows in beta (``eliminate_equalities`` removes them before the device
    sees the problem).  The error tubes have CONSTANT generators here (``:432-436``: W + Zsigma propagated by Acl), their centres
    are affine in (e0, xbar0, v) through ``term_2`` (``:459``; ``term_2 @ Acl`` on a vector is Acl' term_2).
    Zsigma: list of N (centre, generators); Xz, Uz: (centre, generators) of zonotopes.X / .U.
    ze_sum: meaning of the un-vendored ``Ze.sum()`` of ``:451`` -- "radius": row sums of |generators|, "columns": row sums of [c | G]."""
    Acl = np.asarray(Acl, float); Bhat = np.asarray(Bhat, float); K = np.atleast_2d(np.asarray(K, float))
    dA = np.asarray(deltaA, float); dB = np.asarray(deltaB, float)
    n, m = Bhat.shape
    assert len(Zsigma) == N, "Zsigma needs to be a list of zonotopes of length == N, the horizon"
    Xc, XG = np.asarray(Xz[0], float), np.asarray(Xz[1], float).reshape(n, -1)
    Uc, UG = np.asarray(Uz[0], float), np.asarray(Uz[1], float).reshape(m, -1)
    gx, gu = XG.shape[1], UG.shape[1]
    nvv = N * m
    o_bu, o_bx = nvv, nvv + N * gu
    nzb = o_bx + N * gx
    # condensed trajectories: xbar_k = Phi_k xbar0 + Gam_k v ; ubar_k = K xbar_k + v_k ; term_2_k = T0_k xbar0 + Tv_k v
    Phi = np.zeros((N + 1, n, n)); Gam = np.zeros((N + 1, n, nvv)); Phi[0] = np.eye(n)
    for k in range(N):
        Phi[k + 1] = Acl @ Phi[k]; Gam[k + 1] = Acl @ Gam[k]; Gam[k + 1][:, k * m:(k + 1) * m] += Bhat
    Up = np.zeros((N, m, n)); Uv = np.zeros((N, m, nvv))
    for k in range(N):
        Up[k] = K @ Phi[k]; Uv[k] = K @ Gam[k]; Uv[k][:, k * m:(k + 1) * m] += np.eye(m)
    T0 = np.zeros((N + 1, n, n)); Tv = np.zeros((N + 1, n, nvv))
    for k in range(N):
        T0[k + 1] = Acl.T @ T0[k] + dA @ Phi[k] + dB @ Up[k]
        Tv[k + 1] = Acl.T @ Tv[k] + dA @ Gam[k] + dB @ Uv[k]
    # constant part of the tubes: term_1 (:432-436)
    Wc, WG = np.asarray(W_c, float).reshape(n), np.asarray(W_G, float).reshape(n, -1)
    c1 = np.zeros((N, n)); G1 = []
    for k in range(N):
        sc, sG = np.asarray(Zsigma[k][0], float).reshape(n), np.asarray(Zsigma[k][1], float).reshape(n, -1)
        if k == 0:
            c1[0] = Wc + sc; G1.append(np.concatenate([WG, sG], axis=1))
        else:
            c1[k] = Acl @ c1[k - 1] + Wc + sc; G1.append(np.concatenate([Acl @ G1[-1], WG, sG], axis=1))
    gens = [np.zeros((n, 1))] + [np.concatenate([np.zeros((n, 1)), G1[k]], axis=1) for k in range(N)]     # Ze[0 .. N]
    cen_c = np.zeros((N + 1, n)); cen_c[1:] = c1
    # ---- callbacks (:465, :474): loss / constraints on (ubar, xbar[1:]) ---------------------------------------------------
    nsym = nvv + n
    sv, sp = slice(0, nvv), slice(nvv, nsym)
    Cu = np.zeros((nvv, nsym)); Cx = np.zeros(((N + 1) * n, nsym))
    Cu[:, sv] = Uv.reshape(nvv, nvv); Cu[:, sp] = Up.reshape(nvv, n)
    Cx[:, sv] = Gam.reshape((N + 1) * n, nvv); Cx[:, sp] = Phi.reshape((N + 1) * n, n)
    ubar_expr = Affine(Cu, np.zeros(nvv), (N, m)); xbar_expr = Affine(Cx, np.zeros((N + 1) * n), (N + 1, n))
    loss = build_loss(ubar_expr, xbar_expr[1:])
    if loss is None:
        raise Exception("Loss function is not defined or is not convex!")
    loss = cplite.as_convex(loss, nsym)
    if any(w != 0.0 for w, _ in loss.soc):
        raise CpliteError(cplite.SOC_MESSAGE)
    cons = build_constraints(ubar_expr, xbar_expr[1:]) if build_constraints is not None else []
    cons = [] if cons is None else list(cons)
    for idx, c in enumerate(cons):
        if c is None or not isinstance(c, Constraint):
            raise Exception(f"Constraint {idx} is not defined or is not convex.")
    epi_specs = []
    for w, e in loss.ab:
        for r in range(e.size):
            if w != 0.0 and np.any(e.C[r]):
                epi_specs.append((w, [(e.C[r], e.d[r])]))
    for w, e in loss.mx:
        rows_ = [(e.C[r], e.d[r]) for r in range(e.size)]
        if w != 0.0 and any(np.any(c) for c, _ in rows_):
            epi_specs.append((w, rows_))
    s_var0 = nzb
    nz = nzb + len(epi_specs)
    var_names = ([f"v[{k},{j}]" for k in range(N) for j in range(m)] + [f"beta_u[{k},{g}]" for k in range(N) for g in range(gu)]
                 + [f"beta_x[{k},{g}]" for k in range(N) for g in range(gx)] + [f"s[{i}]" for i in range(len(epi_specs))])
    ntheta = 2 * n + N * (2 * n + m)
    blk = 2 * n + m
    ix_c = lambda k, i: 2 * n + k * blk + i
    rows = []
    zt = lambda: np.zeros(ntheta)

    def vrow(coef_v):
        z = np.zeros(nz); z[:nvv] = coef_v
        return z

    for k in range(N):                                                                                  # :417-420
        for g in range(gu):
            z = np.zeros(nz); z[o_bu + k * gu + g] = 1.0; rows.append((f"beta_u[{k},{g}]", z, -1.0, zt(), 1.0, zt()))
        for g in range(gx):
            z = np.zeros(nz); z[o_bx + k * gx + g] = 1.0; rows.append((f"beta_x[{k},{g}]", z, -1.0, zt(), 1.0, zt()))
    for k in range(N):
        for j in range(m):                                                                              # :421
            z = vrow(Uv[k][j]); z[o_bu + k * gu:o_bu + (k + 1) * gu] = -UG[j]
            th = zt(); th[:n] = -Up[k][j]
            rows.append((f"Umem[{k},{j}]", z, Uc[j], th, Uc[j], th.copy()))
        for i in range(n):                                                                              # :422
            z = vrow(Gam[k + 1][i]); z[o_bx + k * gx:o_bx + (k + 1) * gx] = -XG[i]
            th = zt(); th[:n] = -Phi[k + 1][i]
            rows.append((f"Xmem[{k + 1},{i}]", z, Xc[i], th, Xc[i], th.copy()))
    xl, xu = Xc - np.abs(XG).sum(axis=1), Xc + np.abs(XG).sum(axis=1)
    ul, uu = Uc - np.abs(UG).sum(axis=1), Uc + np.abs(UG).sum(axis=1)
    for k in range(N):                                                                                  # :440-449
        radx = np.abs(gens[k]).sum(axis=1); radu = np.abs(K @ gens[k]).sum(axis=1)
        for i in range(n):
            z = vrow(Gam[k][i] + Tv[k][i])
            th = zt(); th[:n] = -(Phi[k][i] + T0[k][i]); th[ix_c(k, i)] -= 1.0
            rows.append((f"X[{k},{i}]", z, xl[i] - cen_c[k, i] + radx[i], th, xu[i] - cen_c[k, i] - radx[i], th.copy()))
        for j in range(m):
            z = vrow(Uv[k][j] + K[j] @ Tv[k])
            th = zt(); th[:n] = -(Up[k][j] + K[j] @ T0[k])
            for i in range(n):
                th[ix_c(k, i)] -= K[j, i]
            kc = K[j] @ cen_c[k]
            rows.append((f"U[{k},{j}]", z, ul[j] - kc + radu[j], th, uu[j] - kc - radu[j], th.copy()))
    # ---- objective ---------------------------------------------------------------------------------------------------------
    P = np.zeros((nz, nz)); q0 = np.zeros(nz); Qt = np.zeros((nz, ntheta))
    r0 = float(loss.const); r1 = loss.lin[sp].copy(); R2 = np.zeros((n, n)); rt = np.zeros(ntheta)
    q0[:nvv] += loss.lin[sv]
    for w, e in loss.sq:
        if w == 0.0:
            continue
        F = np.zeros((e.size, nz)); F[:, :nvv] = e.C[:, sv]; Gp_ = e.C[:, sp]
        P += 2.0 * w * F.T @ F; q0 += 2.0 * w * F.T @ e.d; Qt[:, :n] += 2.0 * w * F.T @ Gp_
        r0 += w * float(e.d @ e.d); r1 += 2.0 * w * Gp_.T @ e.d; R2 += w * Gp_.T @ Gp_
    for w, e in loss.ab:
        for r in range(e.size):
            if not np.any(e.C[r]):
                r0 += w * abs(float(e.d[r]))
    for si, (w, rows_) in enumerate(epi_specs):
        zi = s_var0 + si
        q0[zi] += w
        for Crow, d in rows_:
            zc = vrow(Crow[sv]); e_s = np.zeros(nz); e_s[zi] = 1.0
            th = zt(); th[:n] = Crow[sp]
            rows.append((f"s+[{si}]", e_s - zc, d, th, np.inf, zt()))
            rows.append((f"s-[{si}]", e_s + zc, -d, -th, np.inf, zt()))
    for k in range(N):                                                                                  # :451 regulariser
        if ze_sum == "radius":
            r0 += float(np.abs(gens[k]).sum())
        elif ze_sum == "columns":
            r0 += float(cen_c[k].sum() + gens[k].sum())
            q0[:nvv] += Tv[k].sum(axis=0); r1 += T0[k].sum(axis=0)
            for i in range(n):
                rt[ix_c(k, i)] += 1.0
        else:
            raise ValueError(f"ze_sum={ze_sum!r}")
    for ci, c in enumerate(cons):
        e = c.expr
        for r in range(e.size):
            zc = vrow(e.C[r][sv]); th = zt(); th[:n] = -e.C[r][sp]
            d = -float(e.d[r])
            if c.kind == "<=":
                rows.append((f"user[{ci},{r}]", zc, -np.inf, zt(), d, th))
            elif c.kind == ">=":
                rows.append((f"user[{ci},{r}]", zc, d, th, np.inf, zt()))
            else:
                rows.append((f"user[{ci},{r}]", zc, d, th, d, th.copy()))
    qp_rows = [r for r in rows if np.any(r[1])]
    pr_rows = [r for r in rows if not np.any(r[1])]
    nc = len(qp_rows)
    A = np.array([r[1] for r in qp_rows]).reshape(nc, nz)
    l0 = np.array([r[2] for r in qp_rows]); Lt = np.array([r[3] for r in qp_rows]).reshape(nc, ntheta)
    u0 = np.array([r[4] for r in qp_rows]); Ut = np.array([r[5] for r in qp_rows]).reshape(nc, ntheta)
    f0, Ftl, pl, pu = [], [], [], []
    for name, _, lo_c, lo_t, hi_c, hi_t in pr_rows:              # need lo_c + lo_t theta <= 0 <= hi_c + hi_t theta
        if np.isfinite(lo_c):
            f0.append(lo_c); Ftl.append(lo_t); pl.append(-np.inf); pu.append(0.0)
        if np.isfinite(hi_c):
            f0.append(hi_c); Ftl.append(hi_t); pl.append(0.0); pu.append(np.inf)
    pmax = max(N - 1, 0)
    absCK = np.zeros((max(pmax, 1), n, n)); absKCK = np.zeros((max(pmax, 1), m, n))
    Mp = np.eye(n)
    for j in range(max(pmax, 1)):
        absCK[j] = np.abs(Mp); absKCK[j] = np.abs(K @ Mp); Mp = Acl @ Mp
    tube = TubeConstants(n, m, N, Acl.copy(), np.zeros((n, n)), K.copy(), pmax, absCK, absKCK, np.arange(N, dtype=np.int32))
    qp = ParametricQP(n=n, m=m, N=N, nz=nz, nc=nc, ntheta=ntheta, P=0.5 * (P + P.T), A=A, q0=q0, Qt=Qt, l0=l0, Lt=Lt, u0=u0, Ut=Ut,
                      f0=np.array(f0), Ft=np.array(Ftl).reshape(len(f0), ntheta), pl=np.array(pl), pu=np.array(pu),
                      r0=r0, r1=r1, R2=R2, Phi=Phi.reshape((N + 1) * n, n), Gam=Gam.reshape((N + 1) * n, nvv), tube=tube,
                      row_names=[r[0] for r in qp_rows], var_names=var_names, n_v=nvv, rt=rt if np.any(rt) else None)
    qp.s2 = dict(gens=gens, cen_c=cen_c, T0=T0, Tv=Tv, Up=Up, Uv=Uv)        # for Ze[1] / ubar of the returned solution
    return qp


@dataclass
class Elimination:
    """x = x0 + Xn xbar0 + Z y : the equality rows of the two-sided problem solved for `basic` variables (QR with column
    pivoting); y are the remaining variables in their original order (+ `npad` dummies so that there are at least N m)."""
    x0: np.ndarray          # nz
    Xn: np.ndarray          # nz x n
    Z: np.ndarray           # nz x ny
    free: np.ndarray        # original index of y_k (-1 for the dummies)
    eq_rows: np.ndarray     # rows of the original problem that were equalities
    keep_rows: np.ndarray   # rows of the original problem that stay rows of the reduced one (in that order)


def eliminate_equalities(qp: ParametricQP):
    """User `==` constraints (reference ``tzddpc/tzddpc.py:213-219`` accepts them like any DCP constraint): the interior-point
    kernel works on inequality rows only, so equality rows  E x = f0 + Ft theta  are eliminated on the host at build time.
    Returns (reduced ParametricQP, Elimination) or (qp, None) when there is no equality row."""
    from scipy.linalg import qr, solve_triangular
    fu, fl = np.isfinite(qp.u0), np.isfinite(qp.l0)
    eq = fu & fl & (qp.u0 == qp.l0) & np.all(qp.Ut == qp.Lt, axis=1)
    if not np.any(eq):
        return qp, None
    n, nz, nth = qp.n, qp.nz, qp.ntheta
    E, f0, Ft = qp.A[eq], qp.u0[eq], qp.Ut[eq]
    if np.any(Ft[:, n:]) or np.any(qp.Qt[:, n:]):
        raise NotImplementedError("equality constraints whose right-hand side depends on the tube parameters are not supported")
    Q, R, piv = qr(E, pivoting=True)
    d = np.abs(np.diag(R)) if R.size else np.zeros(0)
    rank = int(np.count_nonzero(d > max(E.shape) * np.finfo(float).eps * (d[0] if d.size else 1.0)))
    rhs = Q.T @ np.hstack([f0[:, None], Ft[:, :n]])
    # dependent equality rows: 0 = rhs_c + rhs_x xbar0 must hold for the problem to be feasible at all -> parameter tests (the
    # reference's solver would report such a problem infeasible at solve time, :374-375 / :496-497)
    dep = [i for i in range(rank, E.shape[0]) if np.abs(rhs[i]).max() > 1e-12 * (1.0 + np.abs(rhs).max())]
    basic, free = piv[:rank], np.sort(piv[rank:])
    R1 = R[:rank, :rank]
    col_of = {int(c): i for i, c in enumerate(piv[rank:])}
    T = solve_triangular(R1, R[:rank, rank:][:, [col_of[int(c)] for c in free]]) if free.size else np.zeros((rank, 0))
    sol = solve_triangular(R1, rhs[:rank])
    npad = max(0, qp.N * qp.m - free.size)
    ny = free.size + npad
    Z = np.zeros((nz, ny)); Z[free, np.arange(free.size)] = 1.0; Z[basic, :free.size] = -T
    x0 = np.zeros(nz); x0[basic] = sol[:, 0]
    Xn = np.zeros((nz, n)); Xn[basic] = sol[:, 1:]
    Xt = np.zeros((nz, nth)); Xt[:, :n] = Xn
    Qn = qp.Qt[:, :n]
    P2 = Z.T @ qp.P @ Z
    q02 = Z.T @ (qp.P @ x0 + qp.q0)
    Qt2 = Z.T @ (qp.P @ Xt + qp.Qt)
    r0 = qp.r0 + 0.5 * x0 @ qp.P @ x0 + qp.q0 @ x0
    r1 = qp.r1 + Xn.T @ (qp.P @ x0 + qp.q0) + Qn.T @ x0
    R2 = qp.R2 + 0.5 * Xn.T @ qp.P @ Xn + Qn.T @ Xn
    # inequality rows; those that lose all their coefficients become parameter tests
    rest = np.nonzero(~eq)[0]
    A2 = qp.A[rest] @ Z
    sh0 = qp.A[rest] @ x0; sht = qp.A[rest] @ Xt
    l02, Lt2 = qp.l0[rest] - sh0, qp.Lt[rest] - sht
    u02, Ut2 = qp.u0[rest] - sh0, qp.Ut[rest] - sht
    scale = np.abs(qp.A[rest]).max(axis=1, initial=0.0) * (1.0 + np.abs(Z).max(initial=0.0))
    A2[np.abs(A2) <= 1e-13 * np.maximum(scale, 1e-300)[:, None]] = 0.0
    live = np.any(A2 != 0.0, axis=1)
    f0p, Ftp, plp, pup = [qp.f0], [qp.Ft], [qp.pl], [qp.pu]
    for i in np.nonzero(~live)[0]:                       # need l0 + Lt theta <= 0 <= u0 + Ut theta
        if np.isfinite(l02[i]):
            f0p.append([l02[i]]); Ftp.append(Lt2[i][None]); plp.append([-np.inf]); pup.append([0.0])
        if np.isfinite(u02[i]):
            f0p.append([u02[i]]); Ftp.append(Ut2[i][None]); plp.append([0.0]); pup.append([np.inf])
    for i in dep:
        row = np.zeros(nth); row[:n] = rhs[i, 1:]
        f0p.append([rhs[i, 0]]); Ftp.append(row[None]); plp.append([-1e-9]); pup.append([1e-9])
    names = [qp.var_names[i] for i in free] + [f"pad[{i}]" for i in range(npad)]
    red = ParametricQP(n=n, m=qp.m, N=qp.N, nz=ny, nc=int(live.sum()), ntheta=nth, P=0.5 * (P2 + P2.T), A=A2[live],
                       q0=q02, Qt=Qt2, l0=l02[live], Lt=Lt2[live], u0=u02[live], Ut=Ut2[live],
                       f0=np.concatenate(f0p), Ft=np.vstack(Ftp), pl=np.concatenate(plp), pu=np.concatenate(pup),
                       r0=float(r0), r1=r1, R2=R2, Phi=qp.Phi, Gam=qp.Gam, tube=qp.tube,
                       row_names=[qp.row_names[i] for i in rest[live]], var_names=names, n_v=qp.n_v, rt=qp.rt)
    free_idx = np.concatenate([free, -np.ones(npad, dtype=free.dtype)])
    return red, Elimination(x0=x0, Xn=Xn, Z=Z, free=free_idx, eq_rows=np.nonzero(eq)[0], keep_rows=rest[live])


def horizon_shift(qp: ParametricQP):
    """Receding-horizon shift of a solution: entry c of the first array is the variable whose previous value is the starting
    guess of variable c one MPC step later (``v[k,i] <- v[k+1,i]``, ``tau[j] <- tau[j+1]`` ...; the last step keeps its own),
    the second array does the same for the (two-sided) rows, i.e. for the multipliers.  Derived from the step index every name
    carries first."""
    import re
    pat = re.compile(r"^([^\[]*)\[(\d+)(.*)$")

    def one(names):
        index = {nm: i for i, nm in enumerate(names)}
        out = np.arange(len(names), dtype=np.int32)
        for i, nm in enumerate(names):
            mt = pat.match(nm)
            if mt:
                out[i] = index.get(f"{mt.group(1)}[{int(mt.group(2)) + 1}{mt.group(3)}", i)
        return out
    return one(qp.var_names), one(qp.row_names)


# ---- host evaluation helpers (used by tests and by the literal Ze[1] export; NOT the hot path) ----

def tube_reference(tube: TubeConstants, e0: np.ndarray):
    """theta tube block for one e0 -- the arithmetic the HIP kernel ``tz_prepare`` performs."""
    n, m, N = tube.n, tube.m, tube.N
    pmax = tube.pmax
    cs = [np.asarray(e0, float)]
    betas = []
    radx = [np.zeros(n)]; radu = [np.zeros(m)]
    for pidx in range(pmax):
        c = cs[-1]
        betas.append(tube.DK @ (np.abs(c) + radx[-1]))
        cs.append(tube.CK @ c)
        rx = np.zeros(n); ru = np.zeros(m)
        for l, b in enumerate(betas):
            rx += tube.absCKpow[pidx - l] @ b
            ru += tube.absKCKpow[pidx - l] @ b
        radx.append(rx); radu.append(ru)
    out = np.zeros(N * (2 * n + m))
    for k in range(N):
        pw = tube.power[k]
        o = k * (2 * n + m)
        out[o:o + n] = cs[pw]; out[o + n:o + 2 * n] = radx[pw]; out[o + 2 * n:o + 2 * n + m] = radu[pw]
    return out


def theta_reference(qp: ParametricQP, xbar0, e0):
    if getattr(qp, "estack", None) is not None:              # literal problem: the e0-only generators evaluated one by one
        from .genstack import evaluate_host
        st = qp.estack
        tb = np.concatenate([np.concatenate(t) for t in evaluate_host(st, np.asarray(e0, float), np.zeros((st.N, st.n + st.m)))[:qp.N]])
    else:
        tb = tube_reference(qp.tube, e0)
    return np.concatenate([np.asarray(xbar0, float), np.abs(np.asarray(xbar0, float)), tb])
