#!/usr/bin/env python3
""" Does the listed frame loop's geometry at one wave per SIMD (16 / 20 modes) pay at every list length?  Frame loop in the batch
    geometry (BILD_NO_LISTED_GEOMETRY=1) against the default, chains of 32 / 40 beads, 10 000 ... 100 000 candidates, k = 4 / 8.
    (profiles/r04_listed_rule.txt: the run that retired the round-3 rule "only while the estimated list fits the chip once".)
        python tools/listed_rule.py """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, helpers as H, bild_amd
from bild_amd import _lib
dev = torch.device('cuda', 0)
T = 1000
for N in (32, 40):
    for n in (10000, 100000):
        for k in (4, 8):
            line = f"N={N} n={n} k={k}:"
            for env in ('BILD_NO_LISTED_GEOMETRY', None):
                for key in ('BILD_NO_LISTED_GEOMETRY',):
                    os.environ.pop(key, None)
                if env:
                    os.environ[env] = '1'
                _lib.config_reload()
                rng = np.random.default_rng(N)
                model = bild_amd.MultiStateRouse(N, 1., 5., d=3, localization_error=0.1)
                traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
                ss, th = H.candidate_profiles(rng, n, k, 2)
                h, ts = model.handle(), model.trajset(traj)
                _lib.logl_st(h, ts, ss[:200], th[:200])
                d_ss, d_th = torch.from_numpy(ss).to(dev), torch.from_numpy(th.astype(np.uint8)).to(dev)
                out = torch.zeros(n, dtype=torch.float64, device=dev)
                go = lambda: _lib.logl_st_device(h, ts, n, k + 1, d_ss.data_ptr(), d_th.data_ptr(), out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
                for _ in range(3):
                    go()
                torch.cuda.synchronize()
                _lib.kernel_timing(True)
                for _ in range(10):
                    go()
                torch.cuda.synchronize()
                _lib.kernel_timing(False)
                ms, c, kn = _lib.kernel_timing_read()
                line += f"  {env or 'default'}: {ms / c * 1e3:8.1f} us"
                del ts, model
            print(line, flush=True)
