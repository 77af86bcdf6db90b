#!/usr/bin/env python3
""" Longer chains (N = 32: 16 modes, transients of > 100 frames): what the caps of the tables cost.
        python tools/n32_probe.py [N] """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, helpers as H, bild_amd
from bild_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, n, k = 1000, 10000, 4
dev = torch.device('cuda', 0)
stream = lambda: torch.cuda.current_stream().cuda_stream
for env in ({}, {'BILD_PAIRS_MAX_GAP': '192'}, {'BILD_PAIRS_MAX_GAP': '255'}, {'BILD_PAIRS_MAX_GAP': '255', 'BILD_STATES_MAX_GAP': '192'},
            {'BILD_STATES_MAX_GAP': '192'}, {'BILD_NO_STATES': '1'}):
    for key, v in env.items():
        os.environ[key] = v
    _lib.config_reload()
    rng = np.random.default_rng(N)
    model = bild_amd.MultiStateRouse(N, 1., 5., d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
    ss, th = H.candidate_profiles(rng, n, k, 2)
    h, ts = model.handle(), model.trajset(traj)
    t0 = time.perf_counter()
    ref = _lib.logl_st(h, ts, ss[:200], th[:200])
    first = time.perf_counter() - t0
    d_ss, d_th = torch.from_numpy(ss).to(dev), torch.from_numpy(th.astype(np.uint8)).to(dev)
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    go = lambda: _lib.logl_st_device(h, ts, n, k + 1, d_ss.data_ptr(), d_th.data_ptr(), out.data_ptr(), stream=stream())
    for _ in range(3):
        go()
    _lib.kernel_timing(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        go()
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / 20
    _lib.kernel_timing(False)
    kms, launches, _ = _lib.kernel_timing_read()
    wms, wl = _lib.kernel_timing_read_walk()
    frames = _lib.frames_run_read(h) / max(launches, 1)
    b, ms = _lib.prefix_info(ts)
    exact = _lib.logl_st(h, ts, ss[:200], th[:200], jump=False)
    print(f"N={N} {env or 'defaults'}: {n / per / 1e6:6.1f} M evals/s, {per * 1e6:6.1f} us/step, frame loop {kms / max(launches, 1) * 1e3:6.1f} us, "
          f"{frames / n:5.1f} frames per candidate, tables {b / 1e6:6.1f} MB built in {ms:5.1f} ms (first call {first * 1e3:5.1f} ms), "
          f"|tables - frame by frame| {np.max(np.abs(ref - exact)):.1e}", flush=True)
    for key in env:
        del os.environ[key]
    del ts, model
_lib.config_reload()
