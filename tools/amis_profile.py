#!/usr/bin/env python3
""" where does the time of a full AMIS step go?  (GPU likelihood vs proposal bookkeeping, on the host or on the device)
    python tools/amis_profile.py [N] [steps] [host|device|auto|fused]   (fused: likelihood + bookkeeping in one native call) """
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else 'auto'
where = {'host': False, 'device': True, 'auto': None, 'fused': True}[mode]
rng = np.random.default_rng(0)
T, k = 1000, 4
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 200), rng=rng)
np.random.seed(1)
sampler = bild_amd.FixedkSampler(traj, model, k=k, N=N, max_fev=10 ** 9, device_bookkeeping=where, fused=(mode == 'fused'))
sampler.step()
t_like = [0.0]
orig = sampler.logL
def timed(ss, thetas):
    t0 = time.perf_counter(); out = orig(ss, thetas); t_like[0] += time.perf_counter() - t0; return out
if mode != 'fused':
    sampler.logL = timed
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(steps):
    sampler.step()
pr.disable()
dt = time.perf_counter() - t0
print(f"N={N}, {'FUSED step, ' if mode == 'fused' else ''}bookkeeping {'on the device' if getattr(sampler._core, 'on_device', False) else 'on the host'}: {steps} AMIS steps in {dt * 1e3:.1f} ms = {dt / steps * 1e3:.2f} ms/step; likelihood (GPU, host buffers) "
      f"{t_like[0] / steps * 1e3:.2f} ms/step; evidence {sampler.evidences[-1]}")
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
