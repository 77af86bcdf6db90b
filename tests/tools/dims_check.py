"""
Throughput and parity by spatial dimension / localization-error pattern (which decides how many mean vectors a
covariance chain carries, hence the lanes per task): 10 000 x T = 1000, 2-state N = 20, modal path.

    python tests/tools/dims_check.py [N]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import _lib
from oracle import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T, k, n = 1000, 4, 10000
for d, err in ((1, 0.1), (2, 0.1), (2, (0.1, 0.2)), (3, 0.1), (3, (0.1, 0.1, 0.3)), (3, (0.1, 0.2, 0.3))):
    rng = np.random.default_rng(1)
    model = bild_amd.MultiStateRouse(N, 1., 5., d=d, localization_error=np.broadcast_to(err, (d,)).copy())
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 200), rng=rng)
    ss, th = H.candidate_profiles(rng, n, k, 2)
    out = model.logL_st_batch(ss, th, traj)
    want = oracle.logl_batch(model.arrays(), model.measurement, model.localization_error, traj[:], H.expand(ss[:64], th[:64], T))
    _lib.kernel_timing(True)
    for _ in range(5):
        model.logL_st_batch(ss, th, traj)
    _lib.kernel_timing(False)
    ms, c, _ = _lib.kernel_timing_read()
    print(f"N={N} d={d} err={err}: kernels {ms / c * 1e3:8.1f} us per batch = {n / (ms / c) / 1e3:6.2f} M evals/s   "
          f"max|diff| vs oracle {np.max(np.abs(out[:64] - want)):.2e}", flush=True)
