"""Golden VECTORS from the only numeric artefacts the reference holds for the hot path: the six stored closed-loop runs of
examples/2.pulley_sim.py:62-103 (examples/results/pulley.xtzddpc.npy, 5 x 201 x 4, and examples/results/xtzddpc.data.npy, 201 x 4:
pulley system, N = 2, x0 = 0, 200 steps, noise W.sample(), un-seeded data set and noise).

The states alone determine everything the loop exchanged with the plant.  The plant is the companion form of
examples/2.pulley_sim.py:39-43 (B = e_1) and W = <0, 0.1 * ones(4)> has ONE generator (:53), so

    x_{t+1} - A x_t = e_1 u_t + ones(4) w_t                                     (:91, w_t = 0.1 beta_t)

is four equations for two unknowns per step: rows 1..3 give w_t three times over (they agree to 4e-16) and row 0 then gives the
applied input u_t = K e_t + v_t[0] (:90).  In all 1 200 stored steps the tube constraints are inactive (the input follows an
affine law exactly), and there

    u_t = K x_t + g_t ,    g_t = v0_t - K xbar_t  ->  g*   geometrically (by 1e-3 every three steps: xbar reaches its fixed point),

so a least-squares fit of u on [x, 1] over the steps >= 30 returns the reference's own gain K of the run (the LMI point its
SDP solver found, tzddpc/utils.py:43-58, solver dependent and not reproducible) to ~1e-13 and g*; g_0 = u_0 = 1 / Bhat[0] (e_0 = 0,
xbar_1[0] = Bhat[0] v0 = 1) is the first entry of the identified input matrix of that run.

DATA only: nothing of the reference's source is read or stored.  Run in the build container (the reference is not on the GPU box):

    python tests/golden/extract_reference_pulley_vectors.py        ->  tests/golden/pulley_reference_vectors.npz
"""
import os

import numpy as np
import scipy.signal as scipysig

SRC5 = "/root/reference/examples/results/pulley.xtzddpc.npy"
SRC1 = "/root/reference/examples/results/xtzddpc.data.npy"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pulley_reference_vectors.npz")
FIT_FROM = 30                                                 # the transient of g_t is below 1e-14 from step 13 on


def recover(x, A):
    """(u, w, consistency residual) of one stored run from x_{t+1} - A x_t = e_1 u_t + ones w_t."""
    d = x[1:] - x[:-1] @ A.T
    w = d[:, 1:].mean(axis=1)
    u = d[:, 0] - w
    resid = np.abs(d - np.outer(u, np.eye(4)[0]) - np.outer(w, np.ones(4))).max()
    return u, w, resid


def main():
    sys_ = scipysig.TransferFunction([0.28261, 0.50666], [1, -1.41833, 1.58939, -1.31608, 0.88642], dt=0.05).to_ss()   # the plant of the runs
    A = np.asarray(sys_.A)
    assert np.array_equal(np.asarray(sys_.B).ravel(), np.eye(4)[0])
    runs = np.concatenate([np.load(SRC5), np.load(SRC1)[None]])
    assert runs.shape == (6, 201, 4) and np.all(runs[:, 0] == 0.0)
    U, Wn, Kf, gs, g, fit_res, rec_res = [], [], [], [], [], [], []
    for x in runs:
        u, w, resid = recover(x, A)
        Phi = np.hstack([x[FIT_FROM:-1], np.ones((200 - FIT_FROM, 1))])
        c = np.linalg.lstsq(Phi, u[FIT_FROM:], rcond=None)[0]
        U.append(u); Wn.append(w); Kf.append(c[:4]); gs.append(c[4]); g.append(u - x[:-1] @ c[:4])
        fit_res.append(np.abs(Phi @ c - u[FIT_FROM:]).max()); rec_res.append(resid)
    U, Wn, Kf, gs, g = map(np.array, (U, Wn, Kf, gs, g))
    assert max(rec_res) < 1e-14 and max(fit_res) < 1e-12 and np.abs(Wn).max() <= 0.1
    np.savez_compressed(
        OUT, source=np.array("rssalessio/TZDDPC examples/results/pulley.xtzddpc.npy + xtzddpc.data.npy (examples/2.pulley_sim.py:62-103)"),
        x=runs, u=U, w=Wn, K=Kf, g_star=gs, g=g, fit_from=np.array(FIT_FROM), fit_residual=np.array(fit_res), recovery_residual=np.array(rec_res))
    print("recovery residual", max(rec_res), "law residual", max(fit_res))
    print("K per run\n", Kf, "\ng*", gs, "\n1/u_0 = Bhat[0]", 1.0 / U[:, 0])


if __name__ == "__main__":
    main()
