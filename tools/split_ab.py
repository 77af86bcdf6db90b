"""
Split launch (table walk + frame loop over the work lists, csrc/walk.hip) against the single launch (BILD_NO_SPLIT):
device time of both kernels per number of switches, candidates resident in HBM, and the largest difference of the results
(must be 0: same numbers added in the same order).    python tools/split_ab.py [n] [T] [k,k,...] [N]
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
N = int(sys.argv[4]) if len(sys.argv) > 4 else 20
model = bild_amd.MultiStateRouse(N, 1., 5., d=3, localization_error=0.1)
traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
h, ts = model.handle(), model.trajset(traj)
dev = torch.device('cuda', 0)
for k in ([int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else (0, 1, 2, 4, 8, 16)):
    if k + 1 > 16:
        continue
    ss, th = H.candidate_profiles(rng, n, k, 2)
    a, b = segments_from_st(ss, th, T)
    da, db = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    outs = {}
    for split in (False, True):
        out = torch.zeros(n, dtype=torch.float64, device=dev)
        def go():
            _lib.logl_segments_device(h, ts, n, k + 1, da.data_ptr(), db.data_ptr(), 0, out.data_ptr(),
                                      stream=torch.cuda.current_stream().cuda_stream, split=split)
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        _lib.kernel_timing(True)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            go()
        t1.record()
        torch.cuda.synchronize()
        _lib.kernel_timing(False)
        ms, c, kn = _lib.kernel_timing_read()
        wms, wc = _lib.kernel_timing_read_walk()
        fr = _lib.frames_run_read(h) / (20.0 * n)
        outs[split] = out.cpu().numpy()
        print(f"n={n} T={T} k={k:2d} {'split ' if split else 'single'}: frame loop {ms / c * 1e3:7.1f} us  walk {wms / max(wc, 1) * 1e3:6.1f} us"
              f"  launch-to-launch {t0.elapsed_time(t1) / 20 * 1e3:7.1f} us   frames per candidate {fr:6.1f}", flush=True)
    print(f"   max |split - single| = {np.max(np.abs(outs[True] - outs[False])):.1e}   NaN: {np.isnan(outs[True]).sum()}", flush=True)
