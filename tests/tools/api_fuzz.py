#!/usr/bin/env python3
"""
Randomised use of the (s, theta) entries: batch sizes around the workgroup size, 0 ... 15 switches, trajectories of 2 ... 300 frames,
2- and 3-state models, one to three dimensions, one or several trajectories (traj_id), sets with and without tables (`expect`),
host entry / device entry / results left on the device, split and single launch.  Every result against the frame-by-frame run
(BILD_NO_PREFIX, which the GPU tests pin to the oracle) to 1e-9, the entries against each other bit for bit.
    python tests/tools/api_fuzz.py [n_cases]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, helpers as H, bild_amd
from bild_amd import _lib

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device('cuda', 0)
worst = 0.0
for c in range(n_cases):
    rng = np.random.default_rng(9100 + c)
    S = int(rng.choice([2, 3]))
    d = int(rng.choice([1, 2, 3]))
    N = int(rng.choice([8, 12, 20]))
    T = int(rng.choice([2, 3, 10, 50, 300]))
    k = int(rng.choice([0, 1, 2, 5, 9, 15]))
    n = int(rng.choice([1, 2, 255, 256, 257, 1000, 3000]))
    n_traj = int(rng.choice([1, 1, 3]))
    expect = None if rng.random() < 0.7 else 5
    model = bild_amd.MultiStateRouse(N, 1., float(rng.choice([0.5, 2., 5.])), d=d, looppositions=H.LOOPS[S], localization_error=float(rng.choice([0.05, 0.2])))
    trajs = [model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, max(1, T // 4)), missing_frames=(0.05 if T > 20 and rng.random() < 0.5 else None), rng=rng)
             for _ in range(n_traj)]
    tid = rng.integers(n_traj, size=n).astype(np.int32) if n_traj > 1 else None
    ts = model.trajset(trajs if n_traj > 1 else trajs[0], expect=expect)
    h = model.handle()
    ss, th = H.candidate_profiles(rng, n, k, S)
    base = _lib.logl_st(h, ts, ss, th, tid, prefix=False)
    got = None
    for rep in range(3):                         # (the tables of a set appear with its first evaluations)
        host = _lib.logl_st(h, ts, ss, th, tid)
        single = _lib.logl_st(h, ts, ss, th, tid, split=False)
        d_ss = torch.from_numpy(np.ascontiguousarray(ss)).to(dev)
        d_th = torch.from_numpy(th.astype(np.uint8)).to(dev)
        d_tid = torch.from_numpy(tid).to(dev) if tid is not None else None
        out = torch.zeros(n, dtype=torch.float64, device=dev)
        _lib.logl_st_device(h, ts, n, k + 1, d_ss.data_ptr(), d_th.data_ptr(), out.data_ptr(), d_traj_id=d_tid.data_ptr() if d_tid is not None else 0)
        torch.cuda.synchronize()
        resident = out.cpu().numpy()
        assert np.array_equal(host, single), (c, rep, 'split vs single')
        assert np.array_equal(host, resident), (c, rep, 'host vs resident')
        if tid is None:
            out.zero_()
            _lib.logl_st_to_device(h, ts, ss, th, out.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(host, out.cpu().numpy()), (c, rep, 'to_device')
        got = host
    dev_ = float(np.max(np.abs(got - base))) if n else 0.0
    worst = max(worst, dev_)
    print(f"case {c:3d}: S={S} N={N} d={d} T={T:3d} k={k:2d} n={n:4d} trajectories={n_traj} tables={'yes' if _lib.prefix_info(ts)[0] > 0 else 'no '}: "
          f"max |default - frame by frame| = {dev_:.2e}; entries agree bit for bit", flush=True)
    assert dev_ < 1e-9, c
    del ts, model
print(f"worst deviation {worst:.2e}")
