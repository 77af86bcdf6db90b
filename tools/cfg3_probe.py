"""
Frame loop of the listed launch on BASELINE configs[3]-like sets, one factor at a time (states, missing frames, T):
device time of the frame loop, frames run, longest task.    python tools/cfg3_probe.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, helpers as H, bild_amd
from bild_amd import _lib
from bild_amd.profiles import segments_from_st

dev = torch.device('cuda', 0)
for S, T, kinds, per, k in ((2, 1000, ['none'], 10000, 4), (2, 2000, ['none'], 10000, 4), (3, 2000, ['none'], 10000, 4),
                            (2, 2000, ['iid'], 10000, 4), (2, 2000, ['bursty'], 10000, 4), (3, 2000, ['bursty'], 10000, 4),
                            (3, 2000, ['none', 'iid', 'bursty', 'none', 'iid', 'bursty'], 5000, 4)):
    rng = np.random.default_rng(33)
    kw = dict(looppositions=H.LOOPS[3]) if S == 3 else {}
    model = bild_amd.MultiStateRouse(20, 1., 5., d=3, localization_error=0.1, **kw)
    trajs = []
    for j, kind in enumerate(kinds):
        miss = H.missing_mask(rng, T, kind)
        trajs.append(model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, T // 5), missing_frames=miss, rng=rng))
    tid = np.repeat(np.arange(len(kinds)), per).astype(np.int32)
    ts = model.trajset(trajs)
    h = model.handle()
    n = len(tid)
    ss, th = H.candidate_profiles(rng, n, k, S)
    _lib.logl_st(h, ts, ss[:500], th[:500], tid[:500])
    dss = torch.from_numpy(np.ascontiguousarray(ss)).to(dev)
    dth = torch.from_numpy(th.astype(np.uint8)).to(dev)
    dtid = torch.from_numpy(tid).to(dev)
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    def go():
        _lib.logl_st_device(h, ts, n, k + 1, dss.data_ptr(), dth.data_ptr(), out.data_ptr(), d_traj_id=dtid.data_ptr(),
                            stream=torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    _lib.kernel_timing(True)
    for _ in range(10):
        go()
    torch.cuda.synchronize()
    _lib.kernel_timing(False)
    ms, c, kn = _lib.kernel_timing_read()
    wms, wc = _lib.kernel_timing_read_walk()
    fr = _lib.frames_run_read(h) / 10.0
    print(f"S={S} T={T} {'+'.join(kinds):40s} n={n}: frame loop {ms / c * 1e3:8.1f} us  walk {wms / max(wc, 1) * 1e3:6.1f} us  frames run {fr:9.0f}", flush=True)
    del ts, model
