"""
The native inference driver on the GPU (csrc/run_host.cpp: bild_run_round -- rows of all trajectories of a round through ONE
bild_logl_st call): against the Python statement of the loop on the same random numbers, and against the oracle.
"""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

TOL = 1e-8


def _trajs(model, rng, n, lo=150, hi=400):
    return [model.trajectory_from_loopingprofile(H.random_profile(rng, int(rng.integers(lo, hi)), 2, 100), rng=rng) for _ in range(n)]


def test_one_trajectory_native_equals_python_on_gpu(built_lib):
    """
    `sample_many([traj])` walks through the random numbers `sample(traj)` does; the likelihoods come out of the same
    tables (a set of one trajectory either way), padded rows included: evidences, log and stream position agree bit for bit.
    """
    import bild_amd
    rng = np.random.default_rng(12)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    for j, traj in enumerate(_trajs(model, rng, 3)):
        np.random.seed(100 + j)
        ref = bild_amd.sample(traj, model, driver='python')
        after_ref = np.random.random_sample()
        np.random.seed(100 + j)
        got = bild_amd.sample_many([traj], model, driver='native')[0]
        after_got = np.random.random_sample()
        np.random.seed(100 + j)
        auto = bild_amd.sample(traj, model)             # 'auto': the native driver for this model and these keywords
        assert np.random.random_sample() == after_ref
        assert np.array_equal(ref.log['k'], auto.log['k']) and np.array_equal(ref.evidence, auto.evidence)
        assert np.array_equal(ref.best_profile()[:], auto.best_profile()[:])
        assert auto.samplers[-1]._adopted                # (it did come out of the native driver)
        assert after_ref == after_got
        assert np.array_equal(ref.log['k'], got.log['k'])
        assert np.array_equal(ref.evidence, got.evidence) and np.array_equal(ref.evidence_se, got.evidence_se)
        assert np.array_equal(ref.log['pk'], got.log['pk'], equal_nan=True) and np.array_equal(ref.log['KLD'], got.log['KLD'], equal_nan=True)
        for sa, sb in zip(ref.samplers, got.samplers):
            assert len(sa.samples) == len(sb.samples)
            for i in (0, len(sa.samples) - 1):
                for key in ('ss', 'thetas', 'logLs'):
                    assert np.array_equal(sa.samples[i][key], sb.samples[i][key]), (sa.k, i, key)
        assert np.array_equal(ref.best_profile()[:], got.best_profile()[:])


def test_many_trajectories_native_on_gpu(built_lib):
    """ 12 trajectories in shared rounds: deterministic, every pooled likelihood is what the oracle says, sensible inference """
    import bild_amd
    from oracle import oracle
    rng = np.random.default_rng(13)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    trajs = _trajs(model, rng, 12)
    np.random.seed(7)
    a = bild_amd.sample_many(trajs, model)
    np.random.seed(7)
    b = bild_amd.sample_many(trajs, model, driver='native')
    for ra, rb in zip(a, b):
        assert np.array_equal(ra.evidence, rb.evidence) and np.array_equal(ra.log['k'], rb.log['k'])
    worst, checked = 0.0, 0
    for r in a:
        T = len(r.traj)
        for s in r.samplers:
            for i in (0, len(s.samples) - 1):
                smp = s.samples[i]
                pick = rng.choice(len(smp['logLs']), min(2, len(smp['logLs'])), replace=False)
                states = H.expand(smp['ss'][pick], smp['thetas'][pick], T)
                want = oracle.logl_batch(model.arrays(), model.measurement, model._get_noise(r.traj), r.traj[:], states)
                worst = max(worst, float(np.max(np.abs(smp['logLs'][pick] - want))))
                checked += len(pick)
    assert checked > 100 and worst < TOL, worst
    assert np.mean([int(r.best_k()) > 0 for r in a]) > 0.5
    # an adopted sampler goes on through the ordinary Python step (GPU likelihood)
    smp = next(s for s in a[0].samplers if not s.exhausted)
    before = len(smp.samples)
    assert smp.step() and len(smp.samples) == before + 1 and np.all(np.isfinite(smp.evidences[-1][:2]))
