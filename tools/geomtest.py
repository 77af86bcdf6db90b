import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import _lib
# usage: [BILD_GEOM=id] python tools/geomtest.py N n [path] [noreduce]   -- kernel time of one geometry
N = int(sys.argv[1]); n = int(sys.argv[2]); path = sys.argv[3] if len(sys.argv) > 3 else 'auto'; T = 1000; k = 4
rng = np.random.default_rng(1)
model = bild_amd.MultiStateRouse(N, 1., 5., d=3, localization_error=0.1, path=path)
if len(sys.argv) > 4:
    a = model.arrays()
    model._handle = _lib.ModelHandle(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], model.measurement, reduce=False)
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 200), rng=rng)
ss, th = H.candidate_profiles(rng, n, k, 2)
model.logL_st_batch(ss, th, traj)
_lib.kernel_timing(True)
for _ in range(3):
    model.logL_st_batch(ss, th, traj)
_lib.kernel_timing(False)
ms, c, _ = _lib.kernel_timing_read()
print(f"N={N} n={n} path={path} geom={os.environ.get('BILD_GEOM', 'auto')}: kernel {ms / c * 1e3:.1f} us")
