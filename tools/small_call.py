#!/usr/bin/env python3
""" A batch of 100 candidates host to host (what one round of the inference driver asks of the likelihood for one trajectory): wall time per
    call and device time of the two kernels, k = 1 ... 5.    python tools/small_call.py """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import _lib
rng = np.random.default_rng(1)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
T = 400
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 100), rng=rng)
h, ts = model.handle(), model.trajset(traj)
for k in (1, 2, 3, 5):
    ss, th = H.candidate_profiles(rng, 100, k, 2)
    _lib.logl_st(h, ts, ss, th)
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        _lib.logl_st(h, ts, ss, th)
    wall = (time.perf_counter() - t0) / n
    _lib.kernel_timing(1)
    for _ in range(50):
        _lib.logl_st(h, ts, ss, th)
    _lib.kernel_timing(False)
    kms, kc, _ = _lib.kernel_timing_read()
    wms, wc = _lib.kernel_timing_read_walk()
    print(f"100 candidates, k={k}: {wall*1e6:6.1f} us per host-to-host call; frame loop {kms/max(kc,1)*1e3:5.1f} us ({kc} launches of 50), walk {wms/max(wc,1)*1e3:5.1f} us", flush=True)
