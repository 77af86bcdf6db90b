#!/usr/bin/env python3
""" cProfile of `bild_amd.sample` (default settings) on a few configs[4]-like trajectories, one after the other: where the host time of
    the adaptive-k loop goes per trajectory.     python tools/sample_profile.py [n_traj] """
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd

n_traj = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(5)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
trajs = [model.trajectory_from_loopingprofile(H.random_profile(rng, int(rng.integers(150, 601)), 2, 120), rng=rng) for _ in range(n_traj)]
np.random.seed(11)
bild_amd.sample(trajs[0], model)   # warm
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = [bild_amd.sample(t, model) for t in trajs]
pr.disable()
dt = time.perf_counter() - t0
steps = sum(len(r.log['k']) for r in res)
print(f"{n_traj} trajectories: {dt:.3f} s, {steps} AMIS steps, {dt / steps * 1e6:.0f} us per step")
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
