"""
BASELINE configs[3] exactly as bench.py builds it, k = 4 and 8: device time of walk and frame loop, frames run, table sizes, the
kernel the frame loop ran in -- to compare switch settings on one box (BILD_STATES_MAX_GAP, BILD_NO_TAIL, BILD_NO_STATES ...).
    python tools/cfg3_k8.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, helpers as H, bild_amd
from bild_amd import _lib

dev = torch.device('cuda', 0)
rng3 = np.random.default_rng(33)
model3 = bild_amd.MultiStateRouse(20, 1., 5., d=3, looppositions=H.LOOPS[3], localization_error=0.1)
T3, per3, kinds = 2000, 5000, ['none', 'iid', 'bursty', 'none', 'iid', 'bursty']
trajs3 = []
for j, kind in enumerate(kinds):
    miss = H.missing_mask(rng3, T3, kind)
    if j % 2 == 1:
        miss = np.union1d(miss, [0])
    trajs3.append(model3.trajectory_from_loopingprofile(H.random_profile(rng3, T3, 3, T3 // 5), missing_frames=miss, rng=rng3))
tid3 = np.repeat(np.arange(len(kinds)), per3).astype(np.int32)
ts3 = model3.trajset(trajs3)
ss3, th3 = H.candidate_profiles(rng3, 500, 4, 3)
_lib.logl_st(model3.handle(), ts3, ss3, th3, tid3[:500])
print("switches:", _lib.config_string() or "(none)", " tables: %.2f GB, built in %.0f ms" % (_lib.prefix_info(ts3)[0] / 1e9, _lib.prefix_info(ts3)[1]), flush=True)
h = model3.handle()
for kk in (4, 8):
    ss, th = H.candidate_profiles(rng3, len(tid3), kk, 3)
    n = len(tid3)
    dss = torch.from_numpy(np.ascontiguousarray(ss)).to(dev)
    dth = torch.from_numpy(th.astype(np.uint8)).to(dev)
    dtid = torch.from_numpy(tid3).to(dev)
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    def go():
        _lib.logl_st_device(h, ts3, n, kk + 1, dss.data_ptr(), dth.data_ptr(), out.data_ptr(), d_traj_id=dtid.data_ptr(),
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
    print(f"  k={kk}: frame loop {ms / c * 1e3:8.1f} us  walk {wms / max(wc, 1) * 1e3:6.1f} us  frames run {fr:9.0f} ({fr / (n * T3):.5f})  kernel {kn}", flush=True)
