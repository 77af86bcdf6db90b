#!/usr/bin/env python3
""" cProfile of the fused adaptive-k loop (`sample_many`) on the BASELINE configs[4] workload (default settings): where the host time goes.
    python tools/config4_profile.py [n_traj] """
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd

n_traj = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(5)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
trajs = []
for j in range(n_traj):
    T = int(rng.integers(150, 601))
    trajs.append(model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 120), rng=rng))
kw = {}
model.logL_segments(np.zeros((1, 1), np.int32), np.zeros((1, 1), np.int32), trajs, np.zeros(1, np.int32))  # upload
np.random.seed(11)
bild_amd.sample_many(trajs[:4], model, return_exceptions=True, **kw)   # warm
np.random.seed(11)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
res = bild_amd.sample_many(trajs, model, return_exceptions=True, **kw)
pr.disable()
print(f"sample_many({n_traj}): {time.perf_counter() - t0:.3f} s")
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
