#!/usr/bin/env python3
"""
Soak test of the table machinery (prefix / transient / pair tables, walk plan, convergence jumps): random models and
trajectories, batches of candidates with 1..15 switches placed to provoke chains (clusters of close switches, switches
at the ends, equal neighbours, empty segments), evaluated with the tables and frame by frame (BILD_NO_PREFIX, which the
GPU tests pin to the oracle).  Prints the largest deviation per configuration; exits 1 above 1e-9.

    python tests/tools/table_soak.py [n_configs] [candidates per config]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import _lib

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
only = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else None      # (a third argument: just these configurations)
worst = 0.0
for c in (only if only is not None else range(n_cfg)):
    rng = np.random.default_rng(7000 + c)
    S = int(rng.choice([2, 2, 3]))
    N = int(rng.choice([10, 16, 20, 24]))
    d = int(rng.choice([1, 2, 3]))
    err = [0.1, [0.1, 0.25, 0.1][:d], 0.3][int(rng.integers(3))]
    model = bild_amd.MultiStateRouse(N, float(rng.choice([0.5, 1, 2])), float(rng.choice([2, 5])), d=d, looppositions=H.LOOPS[S],
                                     localization_error=err)
    n_traj = int(rng.choice([1, 1, 3]))
    Ts = [int(rng.integers(40, 900)) for _ in range(n_traj)]
    trajs = [model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, max(T // 5, 2)),
                                                  missing_frames=[None, 0.05, 0.3][int(rng.integers(3))], rng=rng) for T in Ts]
    K1 = int(rng.integers(2, 17))
    tid = rng.integers(n_traj, size=n).astype(np.int32)
    Tn = np.asarray(Ts)[tid]
    # switch frames: a few cluster centres per candidate, switches scattered closely around them
    centres = (rng.random((n, 3)) * Tn[:, None]).astype(np.int64)
    pick = rng.integers(3, size=(n, K1 - 1))
    spread = rng.choice([2, 8, 30, 80, 400], size=(n, 1))
    starts = np.take_along_axis(centres, pick, axis=1) + rng.integers(-1, 2, size=(n, K1 - 1)) * rng.integers(0, spread, size=(n, K1 - 1))
    starts = np.sort(np.clip(starts, 1, Tn[:, None] + 5), axis=1)          # some beyond the end, some equal (empty segments)
    seg_start = np.concatenate([np.zeros((n, 1), dtype=np.int64), starts], axis=1).astype(np.int32)
    seg_state = np.empty((n, K1), dtype=np.int32)
    seg_state[:, 0] = rng.integers(S, size=n)
    for i in range(1, K1):
        step = rng.integers(0 if rng.random() < 0.2 else 1, S, size=n)       # now and then a boundary that switches nothing
        seg_state[:, i] = (seg_state[:, i - 1] + step) % S
    h, ts = model.handle(), model.trajset(trajs)
    base = _lib.logl_segments(h, ts, seg_start, seg_state, tid, prefix=False)
    fast = _lib.logl_segments(h, ts, seg_start, seg_state, tid)
    again = _lib.logl_segments(h, ts, seg_start[::-1].copy(), seg_state[::-1].copy(), tid[::-1].copy())[::-1]
    single = _lib.logl_segments(h, ts, seg_start, seg_state, tid, split=False)     # one launch instead of walk + frame loop
    nostate = _lib.logl_segments(h, ts, seg_start, seg_state, tid, states=False)   # chains run from their first switch
    dev = float(np.max(np.abs(fast - base)))
    notail = _lib.logl_segments(h, ts, seg_start, seg_state, tid, tail=False)      # transients run until their means have converged too
    dev_notail = float(np.max(np.abs(notail - base)))
    same = bool(np.array_equal(fast, again)) and bool(np.array_equal(fast, single)) and bool(np.array_equal(fast, nostate))
    worst = max(worst, dev)
    print(f"config {c:3d}: S={S} N={N} d={d} trajectories {Ts} K1={K1:2d}: max |tables - frame by frame| = {dev:.2e} on |logL| <= "
          f"{np.max(np.abs(base)):.1e} (without first-order tails {dev_notail:.2e}); order-, split- and state-table-independent: {same}; tables {_lib.prefix_info(ts)[0] / 1e6:.1f} MB", flush=True)
    if not same or not np.all(np.isfinite(fast)):
        print("FAILED"); sys.exit(1)
print(f"worst deviation {worst:.2e}")
sys.exit(0 if worst < 1e-9 else 1)
