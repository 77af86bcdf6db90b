"""
Compare geometries of one padded size: parity against the default geometry and kernel time per launch.

    python tools/geomcmp.py N "id,id,..." "n,n,..." [missing]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import _lib

N = int(sys.argv[1]); ids = sys.argv[2].split(','); ns = [int(x) for x in sys.argv[3].split(',')]
missing = len(sys.argv) > 4
T, k = 1000, 4
rng = np.random.default_rng(1)
model = bild_amd.MultiStateRouse(N, 1., 5., d=3, localization_error=0.1, path='modal')
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 200), rng=rng,
                                            missing_frames=H.missing_mask(rng, T, 'iid') if missing else None)
for n in ns:
    ss, th = H.candidate_profiles(rng, n, k, 2)
    os.environ.pop('BILD_GEOM', None)
    ref = model.logL_st_batch(ss, th, traj)
    for g in ids:
        os.environ['BILD_GEOM'] = g
        out = model.logL_st_batch(ss, th, traj)
        _lib.kernel_timing(True)
        for _ in range(5):
            model.logL_st_batch(ss, th, traj)
        _lib.kernel_timing(False)
        ms, c, _ = _lib.kernel_timing_read()
        print(f"N={N} n={n:7d} geom={g:>3s}: kernel {ms / c * 1e3:8.1f} us   max|diff| vs default {np.max(np.abs(out - ref)):.2e}", flush=True)
