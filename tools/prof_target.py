"""
A fixed number of launches of one batch, for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_target.py ...`:
    python tools/prof_target.py <split|single> [n] [T] [k] [iters] [seam]
candidates resident in HBM (device entry), or with `seam` the host call FixedkSampler.logL uses (bild_logl_st).
Prints the wall time per launch (no events in the stream: what rocprof sees are the kernels alone).
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, helpers as H, bild_amd
from bild_amd import _lib
from bild_amd.profiles import segments_from_st

split = sys.argv[1] != 'single'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
k = int(sys.argv[4]) if len(sys.argv) > 4 else 4
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 50
seam = len(sys.argv) > 6 and sys.argv[6] == 'seam'
rng = np.random.default_rng(2000)
model = bild_amd.MultiStateRouse(20, 1., 5., d=3, localization_error=0.1)
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
h, ts = model.handle(), model.trajset(traj)
dev = torch.device('cuda', 0)
ss, th = H.candidate_profiles(rng, n, k, 2)
a, b = segments_from_st(ss, th, T)
da, db = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
out = torch.zeros(n, dtype=torch.float64, device=dev)
if seam:
    def go():
        return _lib.logl_st(h, ts, ss, th, split=split)
else:
    def go():
        _lib.logl_segments_device(h, ts, n, k + 1, da.data_ptr(), db.data_ptr(), 0, out.data_ptr(),
                                  stream=torch.cuda.current_stream().cuda_stream, split=split)
for _ in range(5):
    go()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    go()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"{'split' if split else 'single'} n={n} T={T} k={k} {'seam' if seam else 'device'}: {dt * 1e6:.1f} us per launch "
      f"({n / dt / 1e6:.1f} M evals/s)", flush=True)
