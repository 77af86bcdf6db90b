"""
The native inference driver (csrc/run_host.cpp, `core.sample_many(driver='native')`) against the Python statement of the
same loop (`core.sample`, itself pinned to the reference's behaviour: tests/test_core.py, tests/test_amis.py).

A run of ONE trajectory consumes the global NumPy stream in the reference's order -- the gamma variates behind
np.random.dirichlet, the uniforms of the traces, the normals of the choice sampler --, so the two must agree BIT FOR
BIT: evidences, their errors, the whole log (k, p(k), KLD, I_la), every pooled sample, and the position the stream is
left at.  CPU only (table likelihoods); the GPU rounds are covered in tests/test_gpu_parity.py.
"""
import copy
import pickle

import numpy as np
import pytest

import bild_amd
from amis_cases import _table
from bild_amd import _lib


class RoutedTables:
    """ table likelihood over several trajectories; a trajectory identifies its table by its first value """

    def __init__(self, tables, n_states=2, transitions=None):
        self.tables = tables
        self.transitions = ~np.eye(n_states, dtype=bool) if transitions is None else np.asarray(transitions, dtype=bool)
        self.nStates, self.d = n_states, 1
        self.launches = 0

    def _one(self, table, seg_start, seg_state):
        T = table.shape[1]
        states = np.empty(T, dtype=int)
        states[:] = seg_state[0]
        for a, s in zip(seg_start[1:], seg_state[1:]):
            if a < T:
                states[a:] = s
        return float(np.sum(table[states, np.arange(T)]))

    def logL_st_batch(self, ss, thetas, traj):
        from bild_amd.profiles import segments_from_st
        a, b = segments_from_st(ss, thetas, len(traj))
        table = self.tables[int(traj[0, 0])]
        return np.array([self._one(table, x, y) for x, y in zip(a, b)])

    def logL(self, profile, traj):
        table = self.tables[int(traj[0, 0])]
        return float(np.sum(table[np.asarray(profile[:], dtype=int), np.arange(len(profile))]))

    def logL_segments(self, seg_start, seg_state, trajs, traj_id):
        self.launches += 1
        return np.array([self._one(self.tables[int(trajs[j][0, 0])], a, b) for a, b, j in zip(seg_start, seg_state, traj_id)])

    def __reduce__(self):
        return (RoutedTables, (self.tables, self.nStates, self.transitions))


def _problem(n, seed=30, n_states=2, length=12, step=3):
    tables = [_table(seed + j, n_states, length + step * j, [4 + j, 9 + j]) for j in range(n)]
    trajs = [bild_amd.Trajectory(np.full((t.shape[1], 1), float(j))) for j, t in enumerate(tables)]
    return tables, trajs


def _same_result(a, b):
    assert np.array_equal(a.evidence, b.evidence) and np.array_equal(a.evidence_se, b.evidence_se)
    assert set(a.log) == set(b.log)
    for key in a.log:
        x, y = a.log[key], b.log[key]
        assert x.shape == y.shape, key
        if x.dtype == object:       # 'I_la' with None entries
            assert all((p is None and q is None) or p == q for p, q in zip(x, y)), key
        else:
            assert np.array_equal(x, y, equal_nan=True), key
    assert len(a.samplers) == len(b.samplers)
    for sa, sb in zip(a.samplers, b.samplers):
        assert sa.k == sb.k and sa.exhausted == sb.exhausted and len(sa.evidences) == len(sb.evidences)
        assert np.array_equal(np.array(sa.evidences), np.array(sb.evidences), equal_nan=True)
        assert ('dirichlet' in sa.__dict__) == ('dirichlet' in sb.__dict__)
        if 'dirichlet' not in sa.__dict__:      # k >= T: the constructor returned before anything else existed (amis.py:641-648)
            continue
        assert len(sa.samples) == len(sb.samples)
        assert sa.logprior == sb.logprior
        for i in range(len(sa.samples)):
            x, y = sa.samples[i], sb.samples[i]
            assert set(x) == set(y)
            for key in x:
                assert np.array_equal(x[key], y[key], equal_nan=True), (sa.k, i, key)
        assert len(sa.parameters) == len(sb.parameters)
        for (a0, l0), (a1, l1) in zip(sa.parameters, sb.parameters):
            assert np.array_equal(a0, a1) and np.array_equal(l0, l1)
    assert a.best_k() == b.best_k() and np.array_equal(a.best_profile()[:], b.best_profile()[:])
    assert np.array_equal(a.log_marginal_posterior(), b.log_marginal_posterior())


CASES = {
    'small': dict(init_runs=3, k_max=4, sampler_kw={'N': 20, 'max_fev': 200, 'max_fcomplete': 30}, choice_kw={'samplesize': 500}),
    'defaults_small_sample': dict(choice_kw={'samplesize': 800}, sampler_kw={'N': 30, 'max_fev': 1500}),
    'exhausts_by_max_fev': dict(init_runs=4, k_max=5, sampler_kw={'N': 25, 'max_fev': 120, 'max_fcomplete': 10}, choice_kw={'samplesize': 400}),
    'no_enumeration_but_k0': dict(init_runs=2, k_max=3, k_lookahead=1, sampler_kw={'N': 15, 'max_fev': 300, 'max_fcomplete': 2}, choice_kw={'samplesize': 300}),
    'margin': dict(dE=1.5, init_runs=2, k_max=6, k_lookahead=3, certainty_in_k=0.9, sampler_kw={'N': 20, 'max_fev': 400}, choice_kw={'samplesize': 600}),
    'one_init_run': dict(init_runs=1, k_max=4, sampler_kw={'N': 20, 'max_fev': 400, 'max_fcomplete': 30}, choice_kw={'samplesize': 300}),
}


@pytest.mark.parametrize('case', list(CASES))
def test_one_trajectory_is_the_python_loop_bit_for_bit(case):
    kw = CASES[case]
    tables, trajs = _problem(3)
    model = RoutedTables(tables)
    for j, traj in enumerate(trajs):
        for seed in (5, 6):
            np.random.seed(seed + 10 * j)
            ref = bild_amd.sample(traj, model, driver='python', **kw)
            after_ref = np.random.random_sample()
            np.random.seed(seed + 10 * j)
            got = bild_amd.sample_many([traj], model, driver='native', **kw)[0]
            after_got = np.random.random_sample()
            _same_result(ref, got)
            assert after_ref == after_got           # the stream was consumed to the same position
            assert got.model is model and all(s.model is model and s.traj is got.traj for s in got.samplers)


def test_three_states_and_restricted_transitions():
    trans = np.array([[0, 1, 1], [1, 0, 0], [1, 1, 0]], dtype=bool)      # state 1 can only go back to 0
    tables = [_table(90 + j, 3, 16 + 2 * j, [5, 11]) for j in range(2)]
    trajs = [bild_amd.Trajectory(np.full((t.shape[1], 1), float(j))) for j, t in enumerate(tables)]
    model = RoutedTables(tables, 3, trans)
    kw = dict(init_runs=3, k_max=4, sampler_kw={'N': 24, 'max_fev': 500, 'max_fcomplete': 40}, choice_kw={'samplesize': 400})
    for j, traj in enumerate(trajs):
        np.random.seed(j)
        ref = bild_amd.sample(traj, model, driver='python', **kw)
        np.random.seed(j)
        got = bild_amd.sample_many([traj], model, driver='native', **kw)[0]
        _same_result(ref, got)


def test_tiny_trajectories_and_refusals():
    """ k >= T samplers, runs in which no sampler ever steps, and the ValueError of an enumeration that may not be made """
    model = RoutedTables([np.log(np.array([[0.6], [0.4]])), np.log(np.array([[0.6, 0.2, 0.7], [0.4, 0.8, 0.3]]))])
    for j, T in ((0, 1), (1, 3)):
        traj = bild_amd.Trajectory(np.full((T, 1), float(j)))
        np.random.seed(1)
        ref = bild_amd.sample(traj, model, k_max=5, driver='python')
        np.random.seed(1)
        got = bild_amd.sample_many([traj], model, driver='native', k_max=5)[0]
        _same_result(ref, got)
        assert all(s.exhausted for s in got.samplers) and len(got.log['k']) == 0
        pickle.loads(pickle.dumps(got))
    # max_fcomplete = 0: k = 0 is enumerated without looking at the limit and CFC.full_sample refuses (bild/amis.py:499-536)
    traj = bild_amd.Trajectory(np.full((3, 1), 1.0))
    with pytest.raises(ValueError, match="Full sample"):
        bild_amd.sample(traj, model, sampler_kw={'max_fcomplete': 0})
    with pytest.raises(ValueError, match="Full sample"):
        bild_amd.sample_many([traj], model, driver='native', sampler_kw={'max_fcomplete': 0})
    out = bild_amd.sample_many([traj], model, driver='native', return_exceptions=True, sampler_kw={'max_fcomplete': 0})
    assert isinstance(out[0], ValueError)


def test_many_trajectories_share_rounds_and_threads_do_not_matter():
    tables, trajs = _problem(7, seed=60, length=14, step=4)
    model = RoutedTables(tables)
    kw = dict(init_runs=3, k_max=4, sampler_kw={'N': 20, 'max_fev': 300, 'max_fcomplete': 30}, choice_kw={'samplesize': 500})
    runs = []
    import os
    for threads in ('1', '4', '4'):
        os.environ['BILD_HOST_THREADS'] = threads
        _lib.config_reload()
        try:
            model.launches = 0
            np.random.seed(123)
            runs.append(bild_amd.sample_many(trajs, model, driver='native', **kw))
            launches = model.launches
        finally:
            del os.environ['BILD_HOST_THREADS']
            _lib.config_reload()
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            _same_result(a, b)
    steps = sum(len(s.samples) for r in runs[0] for s in r.samplers)
    assert launches < steps / 3                              # one likelihood call per ROUND, not per sampler step
    for j, r in enumerate(runs[0]):
        assert len(r.traj) == tables[j].shape[1] and np.all(np.isfinite(r.evidence[:2]))
        lm = r.log_marginal_posterior()
        assert np.allclose(np.sum(np.exp(lm), axis=0), 1.0)
    # the threaded Python driver still exists and is a valid run of the same inference
    np.random.seed(123)
    old = bild_amd.sample_many(trajs, model, driver='python', **kw)
    assert [len(r.samplers) > 0 for r in old] == [True] * len(trajs)


def test_adopted_samplers_go_on_pickle_and_copy():
    """ what the driver hands back are ordinary FixedkSampler objects: they step, pickle and copy like any other """
    tables, trajs = _problem(2, seed=70, length=18)
    model = RoutedTables(tables)
    kw = dict(init_runs=3, k_max=3, sampler_kw={'N': 20, 'max_fev': 10 ** 6, 'max_fcomplete': 30}, choice_kw={'samplesize': 300})
    np.random.seed(9)
    ref = bild_amd.sample(trajs[1], model, driver='python', **kw)
    np.random.seed(9)
    got = bild_amd.sample_many([trajs[1]], model, driver='native', **kw)[0]
    k = max(s.k for s in got.samplers if not s.exhausted)
    a, b = ref.samplers[k], got.samplers[k]
    clones = [pickle.loads(pickle.dumps(b)), copy.deepcopy(b)]
    state = np.random.get_state()
    assert a.step()
    for smp in [b] + clones:
        np.random.set_state(state)
        assert smp.step()
        assert np.array_equal(np.array(smp.evidences), np.array(a.evidences))
        assert np.array_equal(smp.samples[-1]['ss'], a.samples[-1]['ss'])
        assert np.array_equal(smp.samples[0]['log_weights'], a.samples[0]['log_weights'])
        assert len(smp.parameters) == len(a.parameters) and np.array_equal(smp.parameters[-1][0], a.parameters[-1][0])
    back = pickle.loads(pickle.dumps(got))
    assert np.array_equal(back.evidence, got.evidence) and np.array_equal(back.best_profile()[:], got.best_profile()[:])


def test_driver_selection_and_errors():
    tables, trajs = _problem(3, seed=80)
    model = RoutedTables(tables)
    kw = dict(init_runs=2, k_max=3, sampler_kw={'N': 20, 'max_fev': 200, 'max_fcomplete': 30}, choice_kw={'samplesize': 300})
    with pytest.raises(ValueError, match="driver"):
        bild_amd.sample_many(trajs, model, driver='fast', **kw)
    with pytest.raises(ValueError, match="does not apply"):       # device draws are the Python driver's business
        bild_amd.sample_many(trajs, model, driver='native', sampler_kw={'rng': 'device'})
    assert bild_amd.sample_many([], model, **kw) == []

    class Broken(RoutedTables):
        def logL_segments(self, *a):
            raise FloatingPointError("boom")
    with pytest.raises(FloatingPointError):
        bild_amd.sample_many(trajs, Broken(tables), **kw)
    # a plan that was not finished cannot be planned over
    run = _lib.RunHandle([10, 12], model.transitions,
                         dict(init_runs=1, k_lookahead=2, k_max=2, reserved=0, certainty_in_k=0.99, dE=0.0, N=5, concentration_brake=1e-2,
                              polarization_brake=1e-3, max_fev=100, max_fcomplete=30, choice_samplesize=50),
                         [(np.zeros((2, k + 1)), 0.0, 2.0, np.array([[0] * (k + 1), [1] * (k + 1)])) for k in range(3)])
    counts, _ = run.plan()
    assert counts[3] > 0
    with pytest.raises(_lib.BildAmdError, match="not been finished"):
        run.plan()


def test_a_numpy_generator_as_the_source_of_random_numbers():
    tables, trajs = _problem(4, seed=85)
    model = RoutedTables(tables)
    kw = dict(init_runs=3, k_max=4, sampler_kw={'N': 20, 'max_fev': 200, 'max_fcomplete': 30}, choice_kw={'samplesize': 500})
    state = np.random.get_state()[1].copy()
    a = bild_amd.sample_many(trajs, model, rng=np.random.default_rng(3), **kw)
    b = bild_amd.sample_many(trajs, model, rng=np.random.default_rng(3), **kw)
    assert np.array_equal(np.random.get_state()[1], state)         # the global stream was not touched
    for x, y in zip(a, b):
        _same_result(x, y)
    assert all(np.all(np.isfinite(r.evidence[:2])) for r in a)
    with pytest.raises(ValueError, match="native inference driver only"):
        bild_amd.sample_many(trajs, model, driver='python', rng=np.random.default_rng(3), **kw)


def test_sample_takes_the_driver_it_is_told():
    """ `sample(..., driver=)`: 'native' is `sample_many([traj])[0]`, 'auto' stays in Python for a model that is no plain
        MultiStateRouse, a request the native driver cannot serve is refused when it is asked for by name """
    tables, trajs = _problem(2)
    model = RoutedTables(tables)
    kw = CASES['one_init_run']
    np.random.seed(3)
    ref = bild_amd.sample(trajs[0], model, driver='python', **kw)
    np.random.seed(3)
    got = bild_amd.sample(trajs[0], model, driver='native', **kw)
    _same_result(ref, got)
    assert got.samplers[-1]._adopted
    np.random.seed(3)
    auto = bild_amd.sample(trajs[0], model, **kw)
    _same_result(ref, auto)
    assert not getattr(auto.samplers[-1], '_adopted', False)
    with pytest.raises(ValueError):
        bild_amd.sample(trajs[0], model, driver='fastest')
    with pytest.raises(ValueError):
        bild_amd.sample(trajs[0], model, driver='native', show_progress=True)
