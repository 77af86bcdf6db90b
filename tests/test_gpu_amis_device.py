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
