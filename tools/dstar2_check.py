import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, helpers as H, bild_amd
from bild_amd import _lib
rng = np.random.default_rng(20)
T, n, k = 1000, 10000, 4
for errs in (0.1, [0.1, 0.1, 0.25]):
    mod = bild_amd.MultiStateRouse(20, 1., 5., d=3, localization_error=errs)
    tr = mod.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
    ss, th = H.candidate_profiles(rng, n, k, 2)
    h, ts = mod.handle(), mod.trajset(tr)
    a = _lib.logl_st(h, ts, ss, th); b = _lib.logl_st(h, ts, ss, th, split=False)
    dev = torch.device('cuda', 0)
    d1, d2 = torch.from_numpy(ss).to(dev), torch.from_numpy(th.astype(np.uint8)).to(dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    go = lambda: _lib.logl_st_device(h, ts, n, k + 1, d1.data_ptr(), d2.data_ptr(), out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    for _ in range(3): go()
    torch.cuda.synchronize(); _lib.kernel_timing(True)
    for _ in range(20): go()
    torch.cuda.synchronize(); _lib.kernel_timing(False)
    ms, c, _ = _lib.kernel_timing_read(); wms, wc = _lib.kernel_timing_read_walk()
    print(errs, "frame loop", ms / c * 1e3, "us  walk", wms / wc * 1e3, "bit-identical split/single:", np.array_equal(a, b))
