#!/usr/bin/env python3
""" What the first-order tail itself adds to a result, by its tolerance (BILD_TAIL_TOL_BITS: how close the means must be to the
    table's before the rest of a segment is taken from the table): max |with tails - without tails| over batches of candidates,
    several models / data, and the frames run.        python tools/tail_tolerance.py """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import _lib

CASES = [(20, 2, 1000, 4, 'none', 0.1, None), (20, 2, 1000, 8, 'none', 0.1, None), (20, 3, 800, 5, 'bursty', [0.1, 0.1, 0.3], None),
         (32, 2, 1000, 4, 'none', 0.1, None), (24, 3, 700, 10, 'none', 0.1, None), (20, 3, 676, 7, 'none', 0.3, None),
         (20, 2, 800, 5, 'none', 0.1, 'offset'), (20, 2, 800, 5, 'none', 0.1, 'outliers'), (20, 2, 1000, 4, 'none', 0.01, None),
         (16, 2, 500, 6, 'iid', 1.0, None)]
for bits in (24, 22, 20, 18, 16):
    os.environ['BILD_TAIL_TOL_BITS'] = str(bits)
    _lib.config_reload()
    worst_rel, line = 0.0, []
    for N, S, T, k, miss, err, kind in CASES:
        rng = np.random.default_rng(N + T + k)
        model = bild_amd.MultiStateRouse(N, 1, 5, d=3, looppositions=H.LOOPS[S], localization_error=err)
        traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, T // 5), missing_frames=H.missing_mask(rng, T, miss), rng=rng)
        if kind == 'offset':
            traj = bild_amd.Trajectory(traj[:] + 1e3)
        elif kind == 'outliers':
            x = traj[:]
            traj = bild_amd.Trajectory(np.where(rng.random((T, 1)) < 0.01, x + 50 * 0.1 * rng.standard_normal((T, 3)), x))
        ss, th = H.candidate_profiles(rng, 20000, k, S)
        h, ts = model.handle(), model.trajset(traj)
        _lib.logl_st(h, ts, ss[:10], th[:10])
        _lib.frames_run_read(h)
        a = _lib.logl_st(h, ts, ss, th)
        fa = _lib.frames_run_read(h)
        b = _lib.logl_st(h, ts, ss, th, tail=False)
        fb = _lib.frames_run_read(h)
        scale = max(1.0, float(np.max(np.abs(b))) / 1e4)
        dev = float(np.max(np.abs(a - b)))
        worst_rel = max(worst_rel, dev / scale)
        line.append(f"{dev:.1e}({fa / max(fb, 1):.2f})")
        del ts, model
    print(f"bits={bits}: max |tails - no tails| (frames ratio) per case: " + ' '.join(line) + f"   worst / scale {worst_rel:.1e}", flush=True)
