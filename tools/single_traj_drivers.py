#!/usr/bin/env python3
""" One trajectory at a time (the reference's default use): `sample(traj, model)` as the Python loop against the same loop inside
    the native inference driver (`sample_many([traj], model)`: bit-identical, tests/test_gpu_run.py) -- wall time per trajectory
    and per AMIS step, tables built inside the timed call (fresh trajectories).
        python tools/single_traj_drivers.py [n_traj] """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import core

n_traj = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(5)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
mk = lambda: [model.trajectory_from_loopingprofile(H.random_profile(rng, int(rng.integers(150, 601)), 2, 120), rng=rng) for _ in range(n_traj)]
warm = mk()[0]
bild_amd.sample(warm, model); bild_amd.sample_many([warm], model)      # code objects, pools
for name, run in (('python loop ', lambda t: core._sample_python(t, model) if hasattr(core, '_sample_python') else bild_amd.sample(t, model)),
                  ('native driver', lambda t: bild_amd.sample_many([t], model)[0]),
                  ('sample()     ', lambda t: bild_amd.sample(t, model))):
    rng = np.random.default_rng(6)
    trajs = mk()
    np.random.seed(11)
    t0 = time.perf_counter()
    res = [run(t) for t in trajs]
    wall = time.perf_counter() - t0
    steps = sum(len(r.log['k']) for r in res)
    print(f"{name}: {wall / n_traj * 1e3:7.2f} ms per trajectory, {steps} AMIS steps, {wall / steps * 1e6:6.1f} us per step, "
          f"best k {np.bincount([int(r.best_k()) for r in res]).tolist()}", flush=True)
