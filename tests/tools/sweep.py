#!/usr/bin/env python3
"""
Chain-length / localization-error sweep of SURVEY.md section 8d: N in {4, 8, 16, 20, 32}, d* in {1, 2},
10 000 candidate profiles x T = 1000, 2-state, k = 4 on one GPU; beside each the reference Cython kernel
(oracle/_ref) on one host core over a few seconds, and the max |delta logL| between the two on that sample.

    python tests/tools/sweep.py > profiles/r01_sweep.txt          # on the GPU box
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import helpers as H  # noqa: E402
import bild_amd  # noqa: E402
from bild_amd import _lib  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    ref = oracle.load_reference_cython()
    T, n, k = 1000, 10000, 4
    print(f"# N  d*  n_eff  path   GPU evals/s   ms/batch   exec flop/eval   fp64 frac(exec)   CPU ref evals/s (1 core)   speed-up   max|dlogL|")
    for N in (4, 8, 16, 20, 32):
        for errs in ([0.1, 0.1, 0.1], [0.1, 0.1, 0.2]):
            rng = np.random.default_rng(100 * N + len(set(errs)))
            model = bild_amd.MultiStateRouse(N, 1., 5., d=3, localization_error=errs)
            traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
            ss, thetas = H.candidate_profiles(rng, n, k, 2)
            got = model.logL_st_batch(ss, thetas, traj)           # warm-up + results
            h, ts = model.handle(), model.trajset(traj)
            reps = 10
            _lib.kernel_timing(True)
            t0 = time.perf_counter()
            for _ in range(reps):
                model.logL_st_batch(ss, thetas, traj)
            dt = (time.perf_counter() - t0) / reps
            _lib.kernel_timing(False)
            kms, cnt, _ = _lib.kernel_timing_read()
            kms /= max(cnt, 1)
            can, exe = _lib.flop_count(h, ts, n)

            class M:
                pass
            m = M()
            m.models, m.measurement, m.d, m._get_noise = model.models, model.measurement, model.d, model._get_noise
            states = H.expand(ss[:400], thetas[:400], T)
            out, t0 = [], time.perf_counter()
            if ref is not None:
                while time.perf_counter() - t0 < 3.0 and len(out) < len(states):
                    out.append(ref(m, H.ProfileView(states[len(out)]), traj))
            cpu = len(out) / (time.perf_counter() - t0) if out else float('nan')
            diff = np.max(np.abs(got[:len(out)] - np.array(out))) if out else float('nan')
            print(f"{N:3d}  {len(set(errs)):2d}  {h.query(_lib.Q_NEFF):5d}  modal  {n / (kms * 1e-3):12.0f}  {kms:9.3f}  "
                  f"{exe / n:14.0f}  {exe / (kms * 1e-3) / 78.6e12:16.3f}  {cpu:24.1f}  {n / (kms * 1e-3) / cpu:9.0f}  {diff:10.2e}")
            sys.stdout.flush()


if __name__ == '__main__':
    main()
