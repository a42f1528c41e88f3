"""The reference's stored double-integrator figure as vectors (tests/golden/di_reference_figure.npz, written by
tests/golden/extract_reference_di_figure.py from examples/figures/double_integrator.pdf): the 13 states of its TZ-DDPC run
(examples/1.double_integrator_sim.py:75-90: N = 2, x0 = (-5, -2), noise = a random vertex of W per step) and the 13 polygons
Ze[1] + xbar_t it drew (:163-168).  What follows from them, in the order the functions below establish it:

 1. inputs and disturbances.  x_{t+1} - A x_t = B u_t + w_t (:88) with w_t one of the four vertices of W = <0, 0.1 [[1, .5], [.5, 1]]>
    (:51).  The vertices pair up along B = (0.5, 1): each step leaves two candidates, u_t and u_t + 0.2 (one, where |u| <= 1 rules the
    other out).
 2. the run's identified model centre and WHICH candidate it was.  Ze[1] = MdataK <e, 0> + Mdelta <[xbar; v], 0> + W
    (tzddpc/tzddpc.py:172-186 for k = 0) has the centre (Ahat + Bhat K) e, so the centre of the drawn polygon is
    xbar_{t+1} + (Ahat + Bhat K) e_t = Ahat x_t + Bhat u_t -- the gain cancels: 24 linear equations for the six entries of
    [Ahat | Bhat].  Exactly one of the 2^10 candidate sequences fits (1.2e-8; the next best 0.058).
 3. the run's gain and tube magnitudes.  Every polygon is W + a box (its edges: 2 g_1, 2 g_2 of W to the last digit, one horizontal,
    one vertical), the box half-width -- the same in both coordinates: the two rows of |W's generator matrix| have equal sums -- is
        rho_t = dK . |e_t| + dD . |[xbar_t; v_t]|,   v_t = u_t - K e_t,  xbar_{t+1} = Ahat xbar_t + Bhat v_t,  xbar_0 = x_0,
    the single-entry magnitudes of MdataK and Mdelta after reduce(1) (:126-128).  Twelve equations, seven unknowns (K, dK, dD): they
    fit to 3e-9, the parameters are determined to ~1e-7 (Jacobian + the 1.5e-8 rounding of the figure's coordinates).  With them the
    nominal states xbar_t, the errors e_t and the nominal inputs v_t of the reference's run are known.
 4. what the figure's run was.  The top of polygon 4 is at x_2 = 2.000000: the tightened state row of stage 1 was ACTIVE at step 3 with
    the bound 2 (the committed example has 0.95 * 2.5 = 2.375, :52; its shading, drawn by formula, :175, shows that value).  In steps
    4 .. 11 (no tube row active) the reference's input exceeds the solution of the committed `build_problem` by
    0.01 / (2 Bhat'Bhat) = 4.785e-3 in every step: the signature of the loss's `1e-2 * norm(u, 1)` (:27) acting on v, as in
    `build_problem_simplified` (:336), where the committed `build_problem` hands the loss a FREE variable u (:160, :222) on which
    the term is inert.  The figure predates those two lines.  The tests therefore check
      (a) steps 0 .. 3, where tightened tube rows are active (the stage-1 input tube filling all of U in steps 0 and 1, the input
          bound at step 2, the stage-1 state tube at the bound 2 in step 3) and the loss term plays no part: the committed
          formulation reproduces the reference's inputs to 5e-7;
      (b) steps 4 .. 11 with the committed formulation: the predicted offset 0.01 / (2 Bhat'Bhat), to 1e-4;
      (c) all 12 steps, open and closed loop, with the penalty on v: to 1e-4 (the reference's conic solver leaves ~3e-5 in the steps
          whose solution no constraint pins).
This pins, against numbers the reference produced: Ze[1] (centre and radii, i.e. the tube recursion's first step and the reduce(1)
structure), the tightening of the stage-1 input AND state rows in a regime where they are active, the nominal dynamics and the loop.
It does not pin: stages >= 2 of the tube (N = 2), the data -> Mdata step and the gain synthesis (both read off the figure).
"""
import itertools
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
A_TRUE = np.array([[1.0, 1.0], [0.0, 1.0]])                  # examples/1.double_integrator_sim.py:37-38
B_TRUE = np.array([0.5, 1.0])
W_GEN = 0.1 * np.array([[1.0, 0.5], [0.5, 1.0]])            # :51
W_RADIUS = 0.15                                             # interval radius of W, both coordinates
X_LOW = np.array([-8.75, -2.0])                             # x_1: the committed bounds (:52; never active in the run),
X_HIGH = np.array([0.75, 2.0])                              # x_2: the bound the figure's tube touches (point 4 above)
TOL_ACTIVE = 5e-7
TOL_LOOP = 1e-4


def vectors():
    return np.load(os.path.join(GOLD, "di_reference_figure.npz"))


def input_candidates(x):
    """Per step the (u, w) pairs with w a vertex of W, x_{t+1} - A x_t - w parallel to B (1e-6) and |u| <= 1."""
    out = []
    for t in range(len(x) - 1):
        r = x[t + 1] - A_TRUE @ x[t]
        opts = []
        for a, b in itertools.product((1.0, -1.0), repeat=2):
            w = W_GEN @ np.array([a, b]); q = r - w
            if abs(q[0] * B_TRUE[1] - q[1] * B_TRUE[0]) < 1e-6 and abs(q[1]) <= 1.0 + 1e-7:
                opts.append((q[1] / B_TRUE[1], w))
        out.append(opts)
    return out


def polygon_centres_and_halfwidths(g):
    P = g["polygons"]
    return 0.5 * (P.min(axis=1) + P.max(axis=1)), 0.5 * (P.max(axis=1) - P.min(axis=1))


def recover_model_and_inputs(g):
    """-> dict(Ahat, Bhat, u, w, residual, runner_up): least squares of the polygon centres on [x_t, u_t] over every candidate sequence."""
    x = g["x"]
    c, _ = polygon_centres_and_halfwidths(g)
    cands = input_candidates(x)
    fits = []
    for combo in itertools.product(*[range(len(o)) for o in cands]):
        u = np.array([cands[t][k][0] for t, k in enumerate(combo)])
        M = np.hstack([x[:-1], u[:, None]])
        sol = np.linalg.lstsq(M, c[1:], rcond=None)[0]
        fits.append((float(np.abs(M @ sol - c[1:]).max()), combo, sol))
    fits.sort(key=lambda f: f[0])
    err, combo, sol = fits[0]
    return dict(Ahat=sol.T[:, :2].copy(), Bhat=sol.T[:, 2].copy(), u=np.array([cands[t][k][0] for t, k in enumerate(combo)]),
                w=np.array([cands[t][k][1] for t, k in enumerate(combo)]), residual=err, runner_up=fits[1][0],
                ambiguous_steps=sum(len(o) > 1 for o in cands))


def nominal_chain(x, u, Ahat, Bhat, K):
    """(xbar_0..T, e_0..T-1, v_0..T-1) of the loop of examples/1.double_integrator_sim.py:75-90 given the applied inputs."""
    xb = [x[0].copy()]; e = []; v = []
    for t in range(len(u)):
        et = x[t] - xb[t]; vt = u[t] - K @ et
        e.append(et); v.append(vt); xb.append(Ahat @ xb[t] + Bhat * vt)
    return np.array(xb), np.array(e), np.array(v)


def fit_tube_constants(g, m):
    """-> dict(K, dK (2), dD (3), xbar, e, v, residual, sigma): the gain and the single-entry magnitudes that explain the box
    half-widths of the twelve polygons (grid + least squares; the magnitudes must come out non-negative)."""
    from scipy.optimize import least_squares, nnls
    x = g["x"]
    _, half = polygon_centres_and_halfwidths(g)
    rho = half[1:].mean(axis=1) - W_RADIUS

    def design(K):
        xb, e, v = nominal_chain(x, m["u"], m["Ahat"], m["Bhat"], K)
        return np.hstack([np.abs(e), np.abs(xb[:-1]), np.abs(v)[:, None]]), (xb, e, v)

    best = None
    for k1 in np.linspace(-1.5, 0.0, 31):
        for k2 in np.linspace(-2.0, 0.0, 41):
            M, _ = design(np.array([k1, k2]))
            sol, rn = nnls(M, rho)
            if best is None or rn < best[0]:
                best = (rn, np.array([k1, k2]), sol)
    p0 = np.r_[best[1], best[2]]
    r = least_squares(lambda p: design(p[:2])[0] @ p[2:] - rho, p0, xtol=1e-15, ftol=1e-15, gtol=1e-15, x_scale=np.abs(p0) + 1e-3)
    K, mags = r.x[:2], r.x[2:]
    _, (xb, e, v) = design(K)
    cov = np.linalg.inv(r.jac.T @ r.jac) * (1.5e-8) ** 2
    return dict(K=K, dK=mags[:2], dD=mags[2:], xbar=xb, e=e, v=v, residual=float(np.abs(r.fun).max()), sigma=np.sqrt(np.diag(cov)))


def model_matrices(m, f):
    """(Ahat, Bhat (2 x 1), CK, DK (2 x 2), DD (2 x 3), K (1 x 2)) in the form the builders take: both rows of a magnitude matrix
    are equal (point 3 of the module docstring)."""
    Ah = m["Ahat"]; Bh = m["Bhat"].reshape(2, 1); K = f["K"].reshape(1, 2)
    return Ah, Bh, Ah + Bh @ K, np.tile(f["dK"], (2, 1)), np.tile(f["dD"], (2, 1)), K


def l1_offset(m):
    """0.01 / (2 Bhat'Bhat): by how much `1e-2 |v_0|` moves the unconstrained minimiser of |Ahat xbar + Bhat v_0|^2 towards zero."""
    return 0.01 / (2.0 * float(m["Bhat"] @ m["Bhat"]))
