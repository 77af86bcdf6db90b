import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import _lib
rng = np.random.default_rng(0)
model = bild_amd.MultiStateRouse(20, 1., 5., d=3, localization_error=0.1)
for T in (300, 1000):
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
    h, ts = model.handle(), model.trajset(traj)
    for n in (1, 10, 100, 1000):
        for k in (2, 4):
            ss, th = H.candidate_profiles(rng, n, k, 2)
            for _ in range(40): _lib.logl_st(h, ts, ss, th)
            res = {}
            for split in (True, False):
                for _ in range(20): _lib.logl_st(h, ts, ss, th, split=split)
                t0 = time.perf_counter()
                for _ in range(300): _lib.logl_st(h, ts, ss, th, split=split)
                res[split] = (time.perf_counter() - t0) / 300 * 1e6
            print(f"T={T} n={n:5d} k={k}: split {res[True]:6.1f} us   single launch {res[False]:6.1f} us per call", flush=True)
