#!/usr/bin/env python3
""" BASELINE configs[4] (64 trajectories of T ~ U{150..600}, default settings) through `sample_many`: the native inference driver
    against the threaded Python driver, with where the native run's wall time goes (random numbers / rounds / results).
        python tools/config4_native.py [n_traj] [threads ...] """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import _lib, core

n_traj = int(sys.argv[1]) if len(sys.argv) > 1 else 64
thread_counts = [int(v) for v in sys.argv[2:]] or [0]
rng = np.random.default_rng(5)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
trajs = [model.trajectory_from_loopingprofile(H.random_profile(rng, int(rng.integers(150, 601)), 2, 120), rng=rng) for _ in range(n_traj)]
model.logL_segments(np.zeros((1, 1), np.int32), np.zeros((1, 1), np.int32), trajs, np.zeros(1, np.int32))   # upload

# where the time of a native run goes: the three kinds of calls a round makes, timed from outside
acc = {'draws': 0.0, 'round': 0.0, 'plan': 0.0}
for name, key in (('standard_gamma', 'draws'), ('random_sample', 'draws'), ('standard_normal', 'draws')):
    orig = getattr(np.random, name)
    def timed(*a, _o=orig, _k=key, **kw):
        t0 = time.perf_counter()
        try:
            return _o(*a, **kw)
        finally:
            acc[_k] += time.perf_counter() - t0
    setattr(np.random, name, timed)
for name, key in (('round', 'round'), ('plan', 'plan')):
    orig = getattr(_lib.RunHandle, name)
    def timed(self, *a, _o=orig, _k=key, **kw):
        t0 = time.perf_counter()
        try:
            return _o(self, *a, **kw)
        finally:
            acc[_k] += time.perf_counter() - t0
    setattr(_lib.RunHandle, name, timed)

for threads in thread_counts:
    if threads:
        os.environ['BILD_HOST_THREADS'] = str(threads)
    _lib.config_reload()
    for rep in range(3):
        for key in acc:
            acc[key] = 0.0
        np.random.seed(11)
        _lib.kernel_timing(True)
        t0 = time.perf_counter()
        res = bild_amd.sample_many(trajs, model, driver='native', return_exceptions=True)
        wall = time.perf_counter() - t0
        _lib.kernel_timing(False)
        kms, launches, _ = _lib.kernel_timing_read()
        wms, wl = _lib.kernel_timing_read_walk()
        ok = [r for r in res if not isinstance(r, Exception)]
        steps = sum(len(r.log['k']) for r in ok)
        evals = sum(len(smp['logLs']) for r in ok for s in r.samplers for smp in s.samples) if rep == 2 else 0
        print(f"native, BILD_HOST_THREADS={threads or 'default'}: {wall:.3f} s wall, {steps} AMIS steps, {wall / steps * 1e6:.0f} us per step; "
              f"draws {acc['draws'] * 1e3:.0f} ms, rounds {acc['round'] * 1e3:.0f} ms, plans {acc['plan'] * 1e3:.0f} ms, "
              f"rest (results, glue) {(wall - sum(acc.values())) * 1e3:.0f} ms; GPU busy {kms + wms:.1f} ms in {launches + wl} launches"
              + (f"; {evals} evaluations, best k {np.bincount([int(r.best_k()) for r in ok]).tolist()}" if evals else ""))
np.random.seed(11)
t0 = time.perf_counter()
res = bild_amd.sample_many(trajs, model, driver='python', return_exceptions=True)
wall = time.perf_counter() - t0
steps = sum(len(r.log['k']) for r in res if not isinstance(r, Exception))
print(f"python driver (threads as coroutines, fused launches): {wall:.3f} s wall, {steps} AMIS steps, {wall / steps * 1e6:.0f} us per step")
