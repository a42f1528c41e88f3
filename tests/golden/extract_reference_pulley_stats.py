"""Summary statistics of the ONLY numeric artefact the reference holds for the hot path: examples/results/pulley.xtzddpc.npy
(5 runs x 201 states x 4, written by examples/2.pulley_sim.py:62-103: pulley system, N = 2, x0 = 0, 200 closed-loop steps,
noise W.sample() = c + G U(-1, 1), un-seeded, a fresh un-seeded data set per run).  The runs cannot be reproduced (no seed), so
what is kept is the envelope a correct closed loop of the same controller must live in.  DATA only -- nothing of the reference's
source is read or stored.  Run in the build container (the reference is not on the GPU box):

    python tests/golden/extract_reference_pulley_stats.py        ->  tests/golden/pulley_reference_stats.npz
"""
import os

import numpy as np

SRC = "/root/reference/examples/results/pulley.xtzddpc.npy"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pulley_reference_stats.npz")


def main():
    x = np.load(SRC)                                           # (runs, steps + 1, n)
    assert x.shape == (5, 201, 4)
    np.savez_compressed(
        OUT, source=np.array("rssalessio/TZDDPC examples/results/pulley.xtzddpc.npy (examples/2.pulley_sim.py:62-103)"),
        shape=np.array(x.shape), x0=x[:, 0], first_step=x[:, 1],                       # all runs start at 0; state after one step
        state_min=x.min(axis=(0, 1)), state_max=x.max(axis=(0, 1)),                    # global envelope per state
        step_min=x.min(axis=0), step_max=x.max(axis=0), step_mean=x.mean(axis=0),      # (201, 4) envelope over the 5 runs
        tail_mean=x[:, -50:].mean(axis=(0, 1)), tail_std=x[:, -50:].std(axis=(0, 1)),  # last 50 steps, per state
        tail_mean_per_run=x[:, -50:].mean(axis=1), tail_std_per_run=x[:, -50:].std(axis=1))
    g = np.load(OUT)
    print("first step", g["first_step"][:, 0], "tail mean", g["tail_mean"], "tail std", g["tail_std"], "min", g["state_min"], "max", g["state_max"])


if __name__ == "__main__":
    main()
