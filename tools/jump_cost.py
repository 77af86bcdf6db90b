"""
Kernel time of the default path against the number of switches per candidate: t(k) ~ fixed + k * (cost of one switch =
transient frames + convergence checks + basis change + jump).    python tools/jump_cost.py [n] [T]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, helpers as H, bild_amd
from bild_amd import _lib
from bild_amd.profiles import segments_from_st

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(2000)
model = bild_amd.MultiStateRouse(20, 1., 5., d=3, localization_error=0.1)
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
h, ts = model.handle(), model.trajset(traj)
dev = torch.device('cuda', 0)
for k in ([int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else (0, 1, 2, 4, 8, 16)):
    ss, th = H.candidate_profiles(rng, n, k, 2)
    a, b = segments_from_st(ss, th, T)
    da, db = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    def go():
        _lib.logl_segments_device(h, ts, n, k + 1, da.data_ptr(), db.data_ptr(), 0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    _lib.kernel_timing(True)
    for _ in range(20):
        go()
    torch.cuda.synchronize()
    _lib.kernel_timing(False)
    ms, c, kn = _lib.kernel_timing_read()
    fr = _lib.frames_run_read(h) / (20.0 * n)
    print(f"n={n} T={T} k={k:2d}: kernel {ms / c * 1e3:7.1f} us   frames run per candidate {fr:6.1f}", flush=True)
