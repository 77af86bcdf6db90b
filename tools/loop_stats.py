#!/usr/bin/env python3
""" What the adaptive-k loops of BASELINE configs[4] consist of (64 trajectories, default settings): AMIS steps, samplers opened,
    judgements (ChoiceSampler constructions) and their kmax, pool sizes, and where the wall time of `sample_many` goes.
        python tools/loop_stats.py [n_traj] """
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, helpers as H, bild_amd
from bild_amd import core, choicesampler, amis

n_traj = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(5)
model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
trajs = [model.trajectory_from_loopingprofile(H.random_profile(rng, int(rng.integers(150, 601)), 2, 120), rng=rng) for _ in range(n_traj)]
model.logL_segments(np.zeros((1, 1), np.int32), np.zeros((1, 1), np.int32), trajs, np.zeros(1, np.int32))

acc = collections.defaultdict(float)
cnt = collections.Counter()
kmaxes = collections.Counter()


def wrap(cls, name, key):
    orig = getattr(cls, name)

    def timed(self, *a, **kw):
        t0 = time.perf_counter()
        try:
            return orig(self, *a, **kw)
        finally:
            acc[key] += time.perf_counter() - t0
            cnt[key] += 1
    setattr(cls, name, timed)


wrap(choicesampler.ChoiceSampler, '__init__', 'choice.init')
wrap(choicesampler.ChoiceSampler, 'KLD_moreSamples', 'choice.KLD_more')
wrap(choicesampler.ChoiceSampler, 'KLD_omitK', 'choice.KLD_omit')
wrap(amis.FixedkSampler, '__init__', 'sampler.init')
wrap(amis.FixedkSampler, 'step', 'sampler.step')
wrap(amis.Dirichlet, 'sample', 'dirichlet.sample')
orig_init = choicesampler.ChoiceSampler.__init__


def counting_init(self, muhat, *a, **kw):
    kmaxes[len(muhat)] += 1
    return orig_init(self, muhat, *a, **kw)


choicesampler.ChoiceSampler.__init__ = counting_init

np.random.seed(11)
t0 = time.perf_counter()
res = [bild_amd.sample(t, model) for t in trajs]
wall = time.perf_counter() - t0
steps = sum(len(r.log['k']) for r in res)
print(f"{n_traj} trajectories one after the other: {wall:.3f} s, {steps} AMIS steps, {wall / steps * 1e6:.0f} us per step")
for key in sorted(acc):
    print(f"  {key:18s} {cnt[key]:6d} calls  {acc[key] * 1e3:8.1f} ms  {acc[key] / cnt[key] * 1e6:8.1f} us each")
print("  judgements by kmax:", dict(sorted(kmaxes.items())))
print("  samplers opened per trajectory:", np.bincount([len(r.samplers) for r in res]).tolist())
print("  steps per trajectory: min %d mean %.1f max %d" % (min(len(r.log['k']) for r in res), steps / n_traj, max(len(r.log['k']) for r in res)))
pools = [len(s._core) for r in res for s in r.samplers if getattr(s, '_core', None) is not None and not (s.exhausted and not s._sizes)]
print("  final pool sizes: mean %.0f max %d" % (np.mean(pools), max(pools)))
evals = sum(len(smp['logLs']) for r in res for s_ in r.samplers for smp in s_.samples)
print("  likelihood evaluations:", evals)
print("  best k histogram:", np.bincount([int(r.best_k()) for r in res]).tolist())
