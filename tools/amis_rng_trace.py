#!/usr/bin/env python3
""" the AMIS step with draws on the device (opt-in): time per step, the native stages (BILD_AMIS_TRACE=1) and a cProfile of the Python
    side.     python tools/amis_rng_trace.py """
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
rng = np.random.default_rng(0)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, 1000, 2, 200), rng=rng)
np.random.seed(1)
s = bild_amd.FixedkSampler(traj, model, k=4, N=10000, max_fev=10 ** 9, device_bookkeeping=True, fused=True, rng='device', seed=5)
for _ in range(5): s.step()
t0 = time.perf_counter()
for _ in range(40): s.step()
print("ms per step", (time.perf_counter() - t0) / 40 * 1e3)
if not os.environ.get('BILD_AMIS_TRACE'):
    pr = cProfile.Profile(); pr.enable()
    for _ in range(40): s.step()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(10)
