#!/usr/bin/env python3
"""
What the first evaluation on a fresh trajectory set costs (tables are built then), per level of tables, against a later
evaluation: the price `sample(traj, model)` pays once per trajectory.

    python tools/first_call.py [T]           (BILD_NO_PAIRS / BILD_NO_TRANSIENTS / BILD_NO_PREFIX in the environment)
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import _lib
from bild_amd.profiles import segments_from_st

T = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(3)
model = bild_amd.MultiStateRouse(20, 1., 5., d=3, localization_error=0.1)
ss, th = H.candidate_profiles(rng, 100, 3, 2)
a, b = segments_from_st(ss, th, T)
warm = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
model.logL_segments(a, b, warm)                      # device, model upload, code objects
first, later, create = [], [], []
for rep in range(8):
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
    t0 = time.perf_counter()
    ts = model.trajset(traj)
    t1 = time.perf_counter()
    _lib.logl_segments(model.handle(), ts, a, b, None)
    t2 = time.perf_counter()
    _lib.logl_segments(model.handle(), ts, a, b, None)
    t3 = time.perf_counter()
    create.append(t1 - t0); first.append(t2 - t1); later.append(t3 - t2)
flags = ' '.join(k for k in ('BILD_NO_PREFIX', 'BILD_NO_TRANSIENTS', 'BILD_NO_PAIRS') if os.environ.get(k))
print(f"T={T} {flags or 'all tables'}: trajectory set created in {np.median(create) * 1e3:.2f} ms, first evaluation {np.median(first) * 1e3:.2f} ms, "
      f"later ones {np.median(later) * 1e3:.3f} ms; tables {_lib.prefix_info(ts)[0] / 1e6:.1f} MB, device time of the builds {_lib.prefix_info(ts)[1]:.2f} ms")
