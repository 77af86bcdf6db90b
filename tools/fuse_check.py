import os, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, helpers as H, bild_amd
from bild_amd.profiles import segments_from_st
rng = np.random.default_rng(5)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
trajs = []
for j in range(64):
    T = int(rng.integers(150, 601))
    trajs.append(model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 120), rng=rng))
I32 = np.iinfo(np.int32).max
starts, states, tid = [], [], []
K1 = 7
for j, tr in enumerate(trajs):
    k = int(rng.integers(0, 7)); n = 100
    ss, th = H.candidate_profiles(rng, n, k, 2)
    a, b = segments_from_st(ss, th, len(tr))
    if k + 1 < K1:
        a = np.concatenate([a, np.full((n, K1 - k - 1), I32, np.int32)], axis=1)
        b = np.concatenate([b, np.repeat(b[:, -1:], K1 - k - 1, axis=1)], axis=1)
    starts.append(a); states.append(b); tid.append(np.full(n, j, np.int32))
A, B, Tid = np.concatenate(starts), np.concatenate(states), np.concatenate(tid)
fused = model.logL_segments(A, B, trajs, Tid)
print('fused finite:', np.isfinite(fused).all(), 'nan count', np.isnan(fused).sum())
worst = 0
for j, tr in enumerate(trajs):
    sel = Tid == j
    single = model.logL_segments(A[sel], B[sel], [tr], np.zeros(sel.sum(), np.int32))
    worst = max(worst, np.nanmax(np.abs(single - fused[sel])))
    if not np.array_equal(single, fused[sel]):
        bad = np.nonzero(single != fused[sel])[0]
        print('traj', j, 'T', len(tr), 'mismatch at', bad[:5], single[bad[:3]], fused[sel][bad[:3]])
print('worst abs diff fused vs single', worst)
