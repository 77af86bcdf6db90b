#!/usr/bin/env python3
"""
Default-settings inference over a spread of synthetic trajectories (0-6 true switches, T = 80...1000, d = 1...3, two
and three states, 20 to 80 monomers, up to 20 % missing frames, weak to strong signal): does every run complete, and
how good are the answers?

    python tools/robustness_sweep.py [n_traj] [seed] [longest T]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd

n_traj = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
T_hi = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
rng = np.random.default_rng(seed)
total_fail, total, t_all = 0, 0, 0.0
GROUPS = (  # d, localization error, spring constant, monomers, loop positions (one per state)
    (3, 0.1, 5.0, 20, (None, (0, -1))), (2, 0.3, 2.0, 20, (None, (0, -1))), (3, (0.1, 0.1, 0.4), 5.0, 20, (None, (0, -1))),
    (1, 1.0, 1.0, 20, (None, (0, -1))),
    (3, 0.2, 3.0, 24, (None, (0, -1), (0, 12))),      # three states
    (3, 0.1, 5.0, 80, (None, (0, -1))),               # 40 effective modes: the LDS-resident kernel
)
for d, err, kspring, N, loops in GROUPS:
    model = bild_amd.MultiStateRouse(N, 1., kspring, d=d, looppositions=loops, localization_error=err)
    S = len(loops)
    trajs, truths = [], []
    for j in range(n_traj // len(GROUPS)):
        T = int(rng.integers(80, T_hi + 1))
        nsw = int(rng.integers(0, 7))
        cuts = np.sort(rng.choice(np.arange(5, T - 5), size=nsw, replace=False)) if nsw else np.array([], int)
        truth = np.zeros(T, dtype=int)
        s = int(rng.integers(S))
        prev = 0
        for c in list(cuts) + [T]:
            truth[prev:c] = s
            s, prev = int((s + 1 + rng.integers(S - 1)) % S), c
        trajs.append(model.trajectory_from_loopingprofile(bild_amd.Loopingprofile(truth), missing_frames=float(rng.choice([0, 0.05, 0.2])), rng=rng))
        truths.append(truth)
    np.random.seed(seed)
    t0 = time.perf_counter()
    res = bild_amd.sample_many(trajs, model, return_exceptions=True)
    dt = time.perf_counter() - t0
    fails = [r for r in res if isinstance(r, Exception)]
    ok = [(r, t) for r, t in zip(res, truths) if not isinstance(r, Exception)]
    wrong = [float(np.mean(r.best_profile()[:] != t)) for r, t in ok]
    kerr = [int(r.best_k()) - int(np.sum(np.diff(t) != 0)) for r, t in ok]
    finite = all(np.all(np.isfinite(r.log_marginal_posterior()[:, 0]) | True) for r, _ in ok)
    print(f"N={N} S={S} d={d} err={err} k={kspring}: {len(res)} trajectories in {dt:.1f} s, {len(fails)} failed"
          + (f" ({type(fails[0]).__name__}: {fails[0]})" if fails else "")
          + f"; frames wrong: median {np.median(wrong):.3f}, mean {np.mean(wrong):.3f}; best k - true switches: "
          + f"{dict(zip(*np.unique(kerr, return_counts=True)))}", flush=True)
    total_fail += len(fails); total += len(res); t_all += dt
print(f"TOTAL: {total} inferences, {total_fail} failed, {t_all:.1f} s")
