"""
Prefix table and launch order, A/B on one batch: kernel time per launch with (a) every candidate from frame 0 in array
order, (b) prefix table, array order, (c) prefix table, sorted by remaining length, (d) the scheduler's order.

    python tools/prefix_ab.py [n] [T] [k]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
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
_lib.logl_segments(h, ts, a, b)    # builds the tables
sched = _lib.schedule_segments(h, ts, a, b, jump=False)
sched_j = _lib.schedule_segments(h, ts, a, b)
rem = T - np.clip(a[:, 1] if k > 0 else np.full(n, T), 1, T)
plain_sort = np.argsort(-rem, kind='stable').astype(np.int32)
ref = None
for name, order, prefix, jump in (('frame 0, array order', None, False, False), ('prefix, array order', None, True, False),
                                  ('prefix, sorted', plain_sort, True, False), ('prefix, scheduler', sched, True, False),
                                  ('prefix+jumps, array order', None, True, True), ('prefix+jumps, scheduler', sched_j, True, True)):
    do = torch.from_numpy(order).to(dev) if order is not None else None
    def go():
        _lib.logl_segments_device(h, ts, n, k + 1, da.data_ptr(), db.data_ptr(), 0, out.data_ptr(),
                                  stream=torch.cuda.current_stream().cuda_stream, d_order=do.data_ptr() if do is not None else 0,
                                  prefix=prefix, jump=jump)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    _lib.kernel_timing(True)
    for _ in range(20):
        go()
    torch.cuda.synchronize()
    _lib.kernel_timing(False)
    ms, c, kn = _lib.kernel_timing_read()
    res = out.cpu().numpy().copy()
    ref = res if ref is None else ref
    frac = _lib.frames_run_read(h) / (20.0 * n * T)
    print(f"n={n} T={T} k={k}  {name:26s}: kernel {ms / c * 1e3:7.1f} us   frames run {frac:.3f}   max|diff| vs first {np.max(np.abs(res - ref)):.2e}", flush=True)
print("prefix table: %d bytes, built in %.3f ms" % _lib.prefix_info(ts))
