"""
How many frames does a candidate run itself (the rest comes out of the prefix table)?  Distribution over the bench's
batch, and what that means for a wavefront (4 candidates in lockstep: the wave runs as long as its busiest row).

    python tests/tools/frames_hist.py [n] [T] [k]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ctypes
import numpy as np, torch, helpers as H, bild_amd
from bild_amd import _lib
from bild_amd.profiles import segments_from_st

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rng = np.random.default_rng(2000)
model = bild_amd.MultiStateRouse(20, 1., 5., d=3, localization_error=0.1)
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
ss, th = H.candidate_profiles(rng, n, k, 2)
a, b = segments_from_st(ss, th, T)
h, ts = model.handle(), model.trajset(traj)
dev = torch.device('cuda', 0)
da, db = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
out = torch.empty(n, dtype=torch.float64, device=dev)
frames = torch.zeros(n, dtype=torch.int32, device=dev)
_lib.logl_segments_device(h, ts, n, k + 1, da.data_ptr(), db.data_ptr(), 0, out.data_ptr())   # builds the table
torch.cuda.synchronize()
_lib.lib().bild_debug_frames_per_task(ctypes.c_void_p(frames.data_ptr()))
_lib.logl_segments_device(h, ts, n, k + 1, da.data_ptr(), db.data_ptr(), 0, out.data_ptr())
torch.cuda.synchronize()
_lib.lib().bild_debug_frames_per_task(None)
f = frames.cpu().numpy().astype(np.int64)
print(f"n={n} T={T} k={k}: frames run per candidate: mean {f.mean():.1f}, median {np.median(f):.0f}, "
      f"p10 {np.percentile(f, 10):.0f}, p90 {np.percentile(f, 90):.0f}, max {f.max()}")
# per real switch
sw = np.sum((a[:, 1:] < T) & (np.diff(np.concatenate([a[:, :1], a[:, 1:]], axis=1), axis=1) >= 0), axis=1)
print("   mean frames per in-range switch: %.1f" % (f.sum() / max(sw.sum(), 1)))
for name, order in (('array order', np.arange(n)), ('sorted by frames (oracle schedule)', np.argsort(-f, kind='stable'))):
    g = f[order][: n // 4 * 4].reshape(-1, 4)
    wave = g.max(axis=1)
    print(f"   waves of 4, {name}: mean wave length {wave.mean():.1f} (x{wave.mean() / max(f.mean(), 1e-9):.2f} the mean row), longest wave {wave.max()}")
hist = np.bincount(np.minimum(f // 25, 20))
print("   histogram (bins of 25 frames):", hist.tolist())
