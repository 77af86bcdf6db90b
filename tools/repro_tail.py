import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import _lib
from bild_amd.profiles import segments_from_st
err = [0.1, 0.1, 0.3]
rng = np.random.default_rng(22)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=err)
T = 400
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 80), rng=rng)
h, ts = model.handle(), model.trajset(traj)
ss, th = H.candidate_profiles(rng, 6000, 4, 2)
a, b = segments_from_st(ss, th, T)
first = model.logL_segments(a, b, traj)
i = 5123
one = lambda **kw: _lib.logl_segments(h, ts, a[i:i + 1], b[i:i + 1], **kw)[0]
print('first call      ', repr(first[i]))
for kw in ({}, {'states': False}, {'split': False}, {'split': False, 'states': False}, {'tail': False}, {'tail': False, 'states': False}, {'jump': False}):
    print(f'{str(kw):40s}', repr(one(**kw)), repr(_lib.logl_segments(h, ts, a, b, **kw)[i]))
# per chain: which localization-error chain differs?  evaluate the two chains separately through models with one error each
