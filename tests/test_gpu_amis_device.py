"""
The passes of an AMIS step over the pooled samples on the GPU (csrc/amis_device.hip, bild_amis_use_device) against the
host implementation (csrc/amis_host.cpp), which the CPU tests compare step by step with the NumPy formulation and the
reference's own goldens (tests/test_amis.py).  Same per-sample arithmetic (csrc/amis_math.h); sums are formed in a
different (fixed) order, the device's exp / log / log1p differ from the host's libm by an ulp: 1e-10 is the bar.
"""
import pickle

import numpy as np
import pytest

import amis_cases

pytestmark = pytest.mark.gpu


def _steps(case, device, steps=None, **over):
    import bild_amd
    c = dict(amis_cases.CASES[case], **over)
    model = amis_cases.TableModel(c['table'], c['transitions'])
    traj = np.zeros((c['table'].shape[1], 1))
    np.random.seed(c['seed'])
    s = bild_amd.FixedkSampler(traj, model, k=c['k'], N=c['N'], max_fev=c['max_fev'], max_fcomplete=c['max_fcomplete'],
                               device_bookkeeping=device)
    for _ in range(steps or c['steps']):
        if not s.step():
            break
    return s


def _same(a, b, tol=1e-10):
    assert len(a.evidences) == len(b.evidences) and len(a.parameters) == len(b.parameters)
    assert np.allclose(np.array(a.evidences, dtype=float), np.array(b.evidences, dtype=float), rtol=tol, atol=tol, equal_nan=True)
    for (a0, l0), (a1, l1) in zip(a.parameters, b.parameters):
        assert np.allclose(a0, a1, rtol=tol, atol=0) and np.allclose(l0, l1, rtol=0, atol=tol)
    for sa, sb in zip(a.samples, b.samples):
        # (the Dirichlet draws use the fitted concentrations: equal to rounding, not to the bit)
        assert np.array_equal(sa['thetas'], sb['thetas']) and np.allclose(sa['ss'], sb['ss'], rtol=1e-9, atol=1e-12)
        for key in ('logLs', 'logδs', 'cur_log_proposal', 'log_weights'):
            assert np.allclose(sa[key], sb[key], rtol=tol, atol=tol, equal_nan=True), key
    assert np.allclose(a.log_marginal_posterior(), b.log_marginal_posterior(), rtol=0, atol=tol)


@pytest.mark.parametrize('case', ['sampled_k2_T30', 'sampled_k3_3state', 'restricted_transitions'])
def test_device_bookkeeping_equals_host(built_lib, case):
    host, dev = _steps(case, False), _steps(case, True)
    assert dev._core.on_device and not getattr(host._core, 'on_device', False)
    _same(host, dev)


def test_large_batches_go_to_the_device_by_themselves(built_lib):
    """
    N >= 2000 samples per step: the default places the pool in HBM.  8 steps of 3000 samples (pool of 24 000, many blocks
    per pass): the samples the host-side sampler drew are replayed, step by step, into a core on the device -- identical
    inputs, so the comparison does not hinge on two random streams staying in step (a concentration that differs in the
    last bit can flip a rejection inside NumPy's gamma sampler and with it every later draw).
    """
    from bild_amd import _lib
    over = dict(table=amis_cases._table(9, 2, 40, [8, 19, 27, 33]), k=4, N=3000, max_fcomplete=10, max_fev=10 ** 9, steps=8)
    auto = _steps('sampled_k2_T30', None, steps=2, **{k_: v for k_, v in over.items() if k_ != 'steps'})
    assert auto._core.on_device
    small = _steps('sampled_k2_T30', None)                  # the reference's default sizes stay on the host
    assert not getattr(small._core, 'on_device', False)
    host = _steps('sampled_k2_T30', False, **over)
    core = _lib.AmisCore(host.model.transitions, host.parameters[0][0], host.parameters[0][1], host.brakes[0], host.brakes[1],
                         host.logprior)
    core.use_device(True)
    for i, smp in enumerate(host.samples):
        ev = core.step(smp['ss'], smp['thetas'], smp['logLs'])
        assert np.allclose(ev, np.array(host.evidences[i], dtype=float), rtol=1e-10, atol=1e-10)
        a1, l1 = core.params(-1)
        assert np.allclose(a1, host.parameters[i + 1][0], rtol=1e-10, atol=0) and np.allclose(l1, host.parameters[i + 1][1], rtol=0, atol=1e-10)
    for key in core.POOL:
        assert np.allclose(core.pool(key), np.concatenate([smp[key] for smp in host.samples]), rtol=1e-10, atol=1e-10), key


def test_pickled_device_sampler_continues_like_the_host_one(built_lib):
    dev = _steps('sampled_k3_3state', True, steps=3)
    host = _steps('sampled_k3_3state', False, steps=3)
    state = np.random.get_state()
    clone = pickle.loads(pickle.dumps(dev))
    assert clone._core.on_device
    for s in (host, clone):
        np.random.set_state(state)
        assert s.step() and s.step()
    _same(host, clone)


def test_zeros_poles_and_impossible_traces(built_lib):
    """ samples on the boundary of the simplex (x log 0, poles of the density), -inf and NaN likelihoods, traces the
    current proposal cannot produce: the core driven directly, host and device side by side """
    from bild_amd import _lib
    rng = np.random.default_rng(5)
    n, k1, N = 3, 4, 600
    trans = ~np.eye(n, dtype=bool)
    logp0 = np.log(np.full((n, k1), 1. / n))
    cores = []
    for device in (False, True):
        core = _lib.AmisCore(trans, np.ones(k1), logp0, 1e-2, 1e-3, -3.0)
        if device:
            core.use_device(True)
        cores.append(core)
    def batch(step):
        ss = rng.dirichlet(np.ones(k1) * (0.6 if step == 1 else 2.0), size=N)
        ss[::37, 1] = 0.0
        ss[::37] /= ss[::37].sum(axis=1, keepdims=True)
        thetas = np.empty((N, k1), dtype=np.int64)
        thetas[:, 0] = rng.integers(n, size=N)
        for i in range(1, k1):
            thetas[:, i] = (thetas[:, i - 1] + rng.integers(1, n, size=N)) % n
        logLs = -40. * np.sum((ss - np.array([.4, .3, .2, .1])) ** 2, axis=1) + 2. * (thetas[:, 0] == 1)
        logLs[5] = -np.inf
        return ss, thetas, logLs

    def compare():
        for key in cores[0].POOL:
            assert np.allclose(cores[0].pool(key), cores[1].pool(key), rtol=1e-10, atol=1e-10, equal_nan=True), key
        (a0, l0), (a1, l1) = cores[0].params(-1), cores[1].params(-1)
        assert np.allclose(a0, a1, rtol=1e-10, atol=0, equal_nan=True) and np.allclose(l0, l1, rtol=0, atol=1e-10, equal_nan=True)

    for step in range(3):
        ss, thetas, logLs = batch(step)
        ev = [core.step(ss, thetas, logLs) for core in cores]
        assert np.allclose(ev[0], ev[1], rtol=1e-10, atol=1e-10, equal_nan=True)
        compare()
    cores[1].use_device(False)                              # back to the host: state pulled, same next step
    ss = rng.dirichlet(np.ones(k1) * 2.0, size=N)
    thetas = np.tile(np.array([0, 1, 2, 0]), (N, 1))
    logLs = -10. * ss[:, 0]
    ev = [core.step(ss, thetas, logLs) for core in cores]
    assert np.allclose(ev[0], ev[1], rtol=1e-10, atol=1e-10, equal_nan=True)
    cores[1].use_device(True)                               # and up again, with the whole pool
    ss, thetas, logLs = batch(3)
    ev = [core.step(ss, thetas, logLs) for core in cores]
    assert np.allclose(ev[0], ev[1], rtol=1e-10, atol=1e-10, equal_nan=True)
    compare()
    # a NaN likelihood poisons the weights: the slot marginals cannot be inverted, on either side, as in the reference
    ss, thetas, logLs = batch(4)
    logLs[11] = np.nan
    for core in cores:
        with pytest.raises(RuntimeError):
            core.step(ss, thetas, logLs)
    compare()                                               # samples appended on both sides, no new proposal


def _rouse_sampler(seed, fused, N=3000, k=3, T=300, steps=4, **kw):
    import bild_amd
    import helpers as H
    rng = np.random.default_rng(seed)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, T // 5), rng=rng)
    np.random.seed(seed)
    s = bild_amd.FixedkSampler(traj, model, k=k, N=N, max_fev=10 ** 9, max_fcomplete=0, fused=fused, **kw)
    for _ in range(steps):
        s.step()
    return s


def test_fused_step_equals_likelihood_call_plus_bookkeeping(built_lib):
    """
    bild_amis_step_fused: the new samples go up once, as the (s, theta) rows the likelihood kernels read, into the pool in
    HBM; their log-likelihoods are written there; pass A derives on the device what the plain step derives on the host.
    Same kernels, same arithmetic, same order of the sums: every evidence, every proposal and every pooled array equals the
    run that calls `logL` and `bild_amis_step` separately -- bit for bit.  Then a plain step behind fused ones (the host's
    copy of the pool catches up first), a pickled sampler, and MAP / marginals, which read the pool.
    """
    import bild_amd
    fused, plain = _rouse_sampler(5, True), _rouse_sampler(5, False)
    assert fused._core.on_device and fused._fusable() and not plain._fusable()
    assert len(fused._core) == len(plain._core) == 4 * 3000
    assert np.array_equal(np.array(fused.evidences), np.array(plain.evidences))
    for (a0, l0), (a1, l1) in zip(fused.parameters, plain.parameters):
        assert np.array_equal(a0, a1) and np.array_equal(l0, l1)
    for key in fused._core.POOL:
        assert np.array_equal(fused._core.pool(key), plain._core.pool(key)), key
    assert np.array_equal(fused.MAP_profile()[:], plain.MAP_profile()[:])
    assert np.array_equal(fused.log_marginal_posterior(), plain.log_marginal_posterior())
    # a plain step behind fused ones, and a fused one behind that
    state = np.random.get_state()
    fused.fused = False
    fused.step()
    fused.fused = True
    fused.step()
    np.random.set_state(state)
    plain.step()
    plain.step()
    assert np.array_equal(np.array(fused.evidences), np.array(plain.evidences))
    assert np.array_equal(fused._core.pool('log_weights'), plain._core.pool('log_weights'))
    # pickling reads the whole pool back
    clone = pickle.loads(pickle.dumps(fused))
    state = np.random.get_state()
    clone.step()
    np.random.set_state(state)
    fused.step()
    assert np.allclose(np.array(clone.evidences[-1]), np.array(fused.evidences[-1]), rtol=1e-10, atol=1e-10)
    # rows that are no points on the simplex are refused, as by `logL`
    bad = bild_amd.FixedkSampler(fused.traj, fused.model, k=3, N=3000, max_fev=10 ** 9, max_fcomplete=0)
    ss = np.random.dirichlet(np.ones(4), size=3000)
    ss[17, 2] = np.nan
    th = bad._core.sample_traces(np.random.random_sample((4, 3000)))
    with pytest.raises(Exception):
        bad._core.step_fused(bad.model.handle(), bad.model.trajset(bad.traj), ss, th)


def test_device_rng_is_the_same_sampler_in_distribution(built_lib):
    """
    FixedkSampler(rng='device'): the samples of a step are drawn on the GPU (Philox-4x32-10, Marsaglia-Tsang gammas, traces
    slot by slot).  Not the reference's random numbers, so nothing can be compared sample by sample; the sampler must be the
    same in distribution: (i) moments of the very first batch (uniform Dirichlet, uniform traces) against their exact
    values, (ii) the evidence after 6 steps against the NumPy-stream run on the same trajectory, over 20 seeds, within
    the two runs' own standard errors, (iii) reproducible for a seed, different for another one.
    """
    import bild_amd
    import helpers as H
    rng = np.random.default_rng(11)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, 250, 2, 60), rng=rng)
    k, N = 3, 4000

    def run(seed, which, steps=6):
        np.random.seed(seed)
        s = bild_amd.FixedkSampler(traj, model, k=k, N=N, max_fev=10 ** 9, max_fcomplete=0, rng=which, seed=seed)
        for _ in range(steps):
            s.step()
        return s

    first = run(3, 'device', steps=1)
    assert first._device_drawn == N and len(first._core) == N
    ss, thetas = first._pool['ss'], first._pool['thetas']
    assert ss.shape == (N, k + 1) and thetas.shape == (N, k + 1) and np.all(ss >= 0)
    assert np.allclose(ss.sum(axis=1), 1.0, rtol=0, atol=1e-12)
    # Dirichlet(1,1,1,1): mean 1/4, variance 3/80 per coordinate; standard errors of the sample moments at N = 4000
    assert np.all(np.abs(ss.mean(axis=0) - 0.25) < 5 * np.sqrt(3 / 80 / N))
    assert np.all(np.abs(ss.var(axis=0) - 3 / 80) < 0.004)
    assert abs(np.mean(thetas[:, 0]) - 0.5) < 5 * 0.5 / np.sqrt(N)
    assert np.all(thetas[:, 1:] != thetas[:, :-1])          # two states: every slot switches
    # the likelihoods the fused step left in the pool are the likelihoods of exactly these samples
    assert np.array_equal(first._arr['logLs'], model.logL_st_batch(ss, thetas, traj))
    again, other = run(3, 'device', steps=2), run(4, 'device', steps=2)
    assert np.array_equal(np.array(again.evidences[:1]), np.array(first.evidences))
    assert not np.array_equal(np.array(again.evidences), np.array(other.evidences))
    # (i') a proposal that is not uniform, all regimes of the gamma sampler (a < 1: boosted; a > 1), straight from the core:
    # every marginal of Dirichlet(a) is Beta(a_j, sum(a) - a_j) -- Kolmogorov-Smirnov against the exact law, 50 000 draws
    from scipy import stats
    from bild_amd import _lib
    a0 = np.array([0.3, 2.5, 7.0, 40.0])
    logp0 = np.log(np.array([[0.2, 0.5, 0.5, 0.5], [0.8, 0.5, 0.5, 0.5]]))
    core = _lib.AmisCore(model.transitions, a0, logp0, 1e-2, 1e-3, 0.0)
    core.use_device(True)
    core.step_device_rng(model.handle(), model.trajset(traj), 50000, 12345)
    ss, thetas = core.pool_samples()
    for j in range(4):
        pval = stats.kstest(ss[:, j], stats.beta(a0[j], a0.sum() - a0[j]).cdf).pvalue
        assert pval > 1e-4, (j, pval)
    assert abs(thetas[:, 0].mean() - 0.8) < 5 * np.sqrt(0.16 / 50000)
    # (ii) a problem small enough for the exact evidence (every profile evaluated, `fix_exhaustive`).  AMIS evidences after
    # a few steps scatter by more than their reported errors, low rather than high, under EITHER stream (tools/rng_check.py):
    # the two streams are held to each other -- same distribution of the evidence over seeds (rank test) --, and both to
    # the exact value within the scatter.
    short = model.trajectory_from_loopingprofile(H.random_profile(rng, 80, 2, 25), rng=rng)
    exact = bild_amd.FixedkSampler(short, model, k=2, N=100, max_fev=10 ** 6, max_fcomplete=10 ** 5)
    assert exact.exhausted and exact.evidences[-1][1] == 1e-10
    logev = exact.evidences[-1][0]
    got = {'device': [], 'numpy': []}
    for seed in range(12):
        for which in got:
            np.random.seed(500 + seed)
            s_ = bild_amd.FixedkSampler(short, model, k=2, N=4000, max_fev=10 ** 9, max_fcomplete=0, rng=which, seed=500 + seed)
            for _ in range(8):
                s_.step()
            assert (s_._device_drawn > 0) == (which == 'device')
            got[which].append(s_.evidences[-1][0] - logev)
    for which, v in got.items():
        print(f"{which:6s} stream, log-evidence minus the exact {logev:.4f}, 12 seeds:", np.round(v, 3))
    assert stats.mannwhitneyu(got['device'], got['numpy']).pvalue > 1e-3
    for which, v in got.items():
        assert np.max(np.abs(v)) < 3.0, which


def test_device_rng_streams_survive_pickling_and_differ_between_k(built_lib):
    """
    The device-side generator is keyed by (seed, k + 1, index of the sample in the pool): a sampler that was pickled, copied
    or restored must NOT draw its first batches again (round-3 advisor finding: a step counter that restore reset to 0 did),
    and the samplers of one adaptive-k run, which share the user's seed, must not share streams.
    """
    import copy
    import bild_amd
    import helpers as H
    rng = np.random.default_rng(21)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, 300, 2, 60), rng=rng)
    kw = dict(N=2000, max_fev=10 ** 8, max_fcomplete=0, rng='device', seed=99, device_bookkeeping=True)
    smp = bild_amd.FixedkSampler(traj, model, k=3, **kw)
    assert smp.step() and smp.step()
    assert smp._device_drawn == 4000
    clones = [pickle.loads(pickle.dumps(smp)), copy.deepcopy(smp)]
    assert smp.step()
    third = smp.samples[2]['ss'].copy()
    first = smp.samples[0]['ss']
    for clone in clones:
        assert clone.step()
        # the clone goes on exactly like the original: same streams (pool index 4000 ...), same proposal
        assert np.array_equal(clone.samples[2]['ss'], third)
        assert np.array_equal(clone.samples[2]['thetas'], smp.samples[2]['thetas'])
        assert np.allclose(clone.evidences[-1], smp.evidences[-1], rtol=1e-12, atol=0)
    # the uniform first proposal (a = 1) of step 0 and the refit one of step 2 use different streams: were the streams of
    # step 0 reused, the ORDER statistics of the first interval would be the same in both batches (a gamma variate is a
    # monotone function of its uniforms for neighbouring concentrations); with fresh streams the ranks are unrelated
    r0, r2 = np.argsort(np.argsort(first[:, 0])), np.argsort(np.argsort(third[:, 0]))
    assert abs(np.corrcoef(r0, r2)[0, 1]) < 0.1
    # same seed, another k: other streams (k + 1 is part of the key), although both first proposals are uniform
    other = bild_amd.FixedkSampler(traj, model, k=4, **kw)
    assert other.step()
    ra, rb = np.argsort(np.argsort(first[:, 0])), np.argsort(np.argsort(other.samples[0]['ss'][:, 0]))
    assert abs(np.corrcoef(ra, rb)[0, 1]) < 0.1
