#!/usr/bin/env python3
"""
BASELINE configs[4] in miniature: the full adaptive-k AMIS loop (`sample`) on 64 synthetic trajectories of
experimental length (T ~ U{150..600}), run (a) one trajectory after the other, as the reference does, and (b) with
`sample_many`, which fuses the pending AMIS batches of all trajectories into single launches.  Reduced sampler
settings keep the run short; the point is launches per AMIS step and wall time, not the inference quality.

    python tools/config5_demo.py [n_traj]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import _lib

n_traj = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(5)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
trajs = []
for j in range(n_traj):
    T = int(rng.integers(150, 601))
    trajs.append(model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 120), rng=rng))
kw = dict(init_runs=5, k_max=6, sampler_kw={'N': 100, 'max_fev': 3000}, choice_kw={'samplesize': 2000})

model.logL_segments(np.zeros((1, 1), np.int32), np.zeros((1, 1), np.int32), trajs, np.zeros(1, np.int32))  # upload
for name, runner in (('sequential sample()', lambda: [bild_amd.sample(t, model, **kw) for t in trajs]),
                     ('fused sample_many()', lambda: bild_amd.sample_many(trajs, model, return_exceptions=True, **kw))):
    np.random.seed(11)
    _lib.kernel_timing(True)
    t0 = time.perf_counter()
    res = runner()
    dt = time.perf_counter() - t0
    _lib.kernel_timing(False)
    kms, launches, _ = _lib.kernel_timing_read()
    failed = [r for r in res if isinstance(r, Exception)]
    res = [r for r in res if not isinstance(r, Exception)]
    if failed:
        print(f"  {len(failed)} trajectories failed: {failed[0]!r}")
    steps = sum(len(s.samples) for r in res for s in r.samplers)
    evals = sum(len(smp['logLs']) for r in res for s in r.samplers for smp in s.samples)
    ks = [int(r.best_k()) for r in res]
    print(f"{name}: {dt:7.2f} s wall, {steps} AMIS steps, {evals} likelihood evaluations, {launches} kernel launches "
          f"({kms:.0f} ms on the GPU), best k histogram {np.bincount(ks).tolist()}")
    if name.startswith('fused'):
        # what a user does with the results: posterior marginals, best profile, boundary polishing
        from bild_amd import postproc
        t0 = time.perf_counter()
        posts = [r.log_marginal_posterior() for r in res]
        t1 = time.perf_counter()
        best = [r.best_profile() for r in res]
        t2 = time.perf_counter()
        polished = []
        for r, prof in zip(res, best):
            try:
                polished.append(postproc.optimize_boundary(prof, r.traj, model))
            except postproc.BoundaryEliminationError:
                polished.append(prof)
        t3 = time.perf_counter()
        moved = sum(int(np.sum(a[:] != b[:])) for a, b in zip(best, polished))
        print(f"post-processing of {len(res)} results: log_marginal_posterior {t1 - t0:.2f} s, best_profile {t2 - t1:.3f} s, "
              f"optimize_boundary {t3 - t2:.2f} s ({moved} frames moved)")
