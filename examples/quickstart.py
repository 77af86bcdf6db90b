#!/usr/bin/env python3
"""
BILD on an MI355X in a dozen lines: simulate two-locus trajectories from a looping profile, infer the
profiles back.  Same calls as with the reference package (`bild.sample`, `bild.postproc.optimize_boundary`),
plus `sample_many`, which fuses the likelihood batches of all trajectories into single GPU launches.

    python examples/quickstart.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bild_amd
from bild_amd import postproc

rng = np.random.default_rng(1)
model = bild_amd.MultiStateRouse(N=20, D=1., k=5., d=3, localization_error=0.1)   # state 0: free chain, 1: looped

# ground truth: looped between frames 120-260 and 400-470
truth = np.zeros(600, dtype=int)
truth[120:260] = 1
truth[400:470] = 1
trajs = [model.trajectory_from_loopingprofile(bild_amd.Loopingprofile(truth), missing_frames=0.05, rng=rng) for _ in range(8)]

np.random.seed(0)
results = bild_amd.sample_many(trajs, model)              # one adaptive-k AMIS inference per trajectory, fused launches
for j, res in enumerate(results):
    profile = res.best_profile()
    try:
        profile = postproc.optimize_boundary(profile, res.traj, model)
    except postproc.BoundaryEliminationError:
        pass
    posterior = np.exp(res.log_marginal_posterior())      # (states, frames)
    wrong = int(np.sum(profile[:] != truth))
    print(f"trajectory {j}: best k = {res.best_k()}, switches at {np.nonzero(np.diff(profile[:]))[0] + 1}, "
          f"{wrong} of {len(truth)} frames differ from the truth, mean P(looped | data) inside the loops "
          f"{posterior[1, truth == 1].mean():.3f}")
