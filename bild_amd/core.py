"""
The adaptive-k outer loop: `sample` and `SamplingResults`.

Counterpart of reference bild/core.py:22-372.  Control flow only:
which k gets the next AMIS step is decided on the host from the evidence curve
(`ChoiceSampler`); every likelihood evaluation goes through `FixedkSampler.logL`, i.e. one
GPU launch per AMIS step with the GPU-backed model.

`sample_many` (no reference counterpart) runs the loop for many trajectories concurrently and
fuses the pending AMIS batches of all of them into single launches (`batching.BatchingModel`).
"""
import numpy as np
from scipy.special import logsumexp

from .amis import FixedkSampler
from .choicesampler import ChoiceSampler
from .trajectory import make_trajectory

__all__ = ['sample', 'sample_many', 'SamplingResults']


def sample(traj, model,
           dE=0,
           init_runs=20,
           certainty_in_k=0.99,
           k_lookahead=2,
           k_max=20,
           sampler_kw={},
           choice_kw={},
           show_progress=False,
           driver='auto',
           ):
    """
    Run BILD on one trajectory (reference bild/core.py:22-236): AMIS samplers for k = 0, 1, ...
    are created on demand; after each step the choice distribution p(k) decides whether to
    refine an existing k, open the next one (look-ahead rule), or stop
    (``max p(k) >= certainty_in_k``).

    Parameters and defaults are those of the reference.  Returns `SamplingResults` -- also on
    ``KeyboardInterrupt``, with whatever has been sampled so far.

    ``driver`` (not in the reference): ``'python'`` runs the loop below, step by step in Python; ``'native'`` runs the same
    loop inside the native inference driver (`sample_many`: same random numbers in the same order, same result bit for bit
    -- tests/test_run.py, tests/test_gpu_run.py -- at half to three quarters of the wall time per AMIS step); ``'auto'`` takes
    the native driver where it applies without a change of behaviour: a plain `MultiStateRouse` model, the keywords
    `sample_many` lists, no progress bar.
    """
    if driver not in ('auto', 'native', 'python'):
        raise ValueError("driver must be 'auto', 'native' or 'python'")
    if driver != 'python':
        plan, why = _native_plan(model, dict(dE=dE, init_runs=init_runs, certainty_in_k=certainty_in_k, k_lookahead=k_lookahead, k_max=k_max,
                                             sampler_kw=sampler_kw, choice_kw=choice_kw, show_progress=show_progress))
        if plan is not None and (driver == 'native' or plan['mode'] == 'gpu'):
            return _sample_many_native([traj], model, False, plan, interruptible=True)[0]
        if driver == 'native':
            raise ValueError("the native inference driver does not apply: " + why)
    return _sample_python(traj, model, dE, init_runs, certainty_in_k, k_lookahead, k_max, sampler_kw, choice_kw, show_progress)


def _sample_python(traj, model, dE=0, init_runs=20, certainty_in_k=0.99, k_lookahead=2, k_max=20, sampler_kw={}, choice_kw={},
                   show_progress=False):
    """ `sample`, the loop in Python (what the reference runs) """
    run = _AdaptiveRun(make_trajectory(traj), model, dE, init_runs, certainty_in_k, k_lookahead, k_max,
                       dict(sampler_kw), dict(choice_kw), _progress_bar(show_progress))
    try:
        run.loop()
    except KeyboardInterrupt:  # pragma: no cover
        pass
    return SamplingResults(run.traj, model, dE, run.samplers, run.log_columns())


class _AdaptiveRun:
    """
    The adaptive-k loop of `sample` as a small state machine.

    State: the samplers opened so far (index = k), one log row per AMIS step that was really taken, and `target`, the k
    the next round works on (``len(samplers)`` = open a new one).  A round is `work` (take the step(s)) followed by
    `judge` (rate the evidence curve, annotate the newest row, pick the next target) and `goes_on` (the stop rules).  The
    decisions are those of reference bild/core.py:130-227; the random numbers are consumed in the same order (one
    `ChoiceSampler` per round that took a step), so a seeded run reproduces the reference's sequence of k.
    """

    def __init__(self, traj, model, dE, init_runs, certainty_in_k, k_lookahead, k_max, sampler_kw, choice_kw, progress):
        self.traj, self.model, self.dE = traj, model, dE
        self.init_runs, self.certainty, self.lookahead, self.k_max = init_runs, certainty_in_k, k_lookahead, k_max
        self.sampler_kw, self.choice_kw, self.progress = sampler_kw, choice_kw, progress
        self.samplers = []
        self.rows = []          # {'k', 'pk', 'KLD', 'I_la'} per step taken; only the newest row of a round is annotated
        self.target = 0

    # -- bookkeeping -----------------------------------------------------------------------------------------------
    def log_columns(self):
        return {key: [row[key] for row in self.rows] for key in ('k', 'pk', 'KLD', 'I_la')}

    def _step(self, k):
        """ one AMIS step of sampler k; False when that sampler is exhausted (nothing happens then) """
        if not self.samplers[k].step():
            return False
        self.progress.update()
        self.rows.append({'k': k, 'pk': None, 'KLD': None, 'I_la': None})
        return True

    def work(self):
        """ the round's sampling: a new sampler gets `init_runs` steps at once, an existing one a single step """
        k = self.target
        if k > len(self.samplers):  # pragma: no cover
            raise RuntimeError("Trying to sample outside of existing range; this is a bug")
        if k < len(self.samplers):
            return self._step(k)
        self.samplers.append(FixedkSampler(self.traj, self.model, k=k, **self.sampler_kw))
        return any([self._step(k) for _ in range(self.init_runs)])

    # -- decisions -------------------------------------------------------------------------------------------------
    def judge(self):
        """ rate the evidence curve after a round that took a step; annotates the newest row and sets the next target """
        opened = len(self.samplers)
        may_open = opened <= self.k_max
        last = [s.evidences[-1] for s in self.samplers]
        budget = [np.inf if s.exhausted else len(s.samples) for s in self.samplers]
        cs = ChoiceSampler(np.array([ev[0] for ev in last]), np.array([ev[1] for ev in last]) ** 2, np.array(budget),
                           self.dE, **self.choice_kw)
        row = self.rows[-1]
        row['pk'] = cs.n0 / cs.samplesize
        if opened <= self.lookahead and may_open:
            # every sampler still lies inside the look-ahead window: whatever is behind it matters infinitely much
            row['I_la'] = np.inf
            self.target = opened
            return
        gain = cs.KLD_moreSamples()
        window = np.arange(opened - self.lookahead, opened)
        row['KLD'] = gain
        row['I_la'] = cs.KLD_omitK(window) if opened > self.lookahead else np.inf
        refine = int(np.argmax(gain))
        self.target = opened if (may_open and row['I_la'] > gain[refine]) else refine

    def goes_on(self, stepped):
        k = self.target
        if k == len(self.samplers):
            # a new k takes precedence over the certainty criterion.  (When no sampler has taken a single step
            # yet -- every k so far was enumerated exhaustively or has k >= T, i.e. a trajectory of a few frames
            # -- the reference keeps opening samplers without looking at k_max, forever; here k_max ends it.)
            return k <= self.k_max
        if not stepped and self.samplers[k].exhausted:
            # nothing was sampled in this round and the next candidate cannot sample either: the reference
            # would spin on it; there is nothing left to learn
            return False
        newest = self.rows[-1]
        if np.max(newest['pk']) >= self.certainty:
            return False
        # no information left to gain when all relevant samplers are exhausted
        return newest['KLD'] is None or newest['KLD'][k] > 0

    def loop(self):
        while True:
            stepped = self.work()
            if stepped:
                self.judge()
            elif not self.rows:
                self.target = len(self.samplers)
            else:
                self.target = self.rows[-1]['k']
            if not self.goes_on(stepped):
                break
        self.progress.close()


def sample_many(trajs, model, driver='auto', return_exceptions=False, rng=None, **kwargs):
    """
    `sample` for a list of trajectories, run concurrently: one ROUND advances the adaptive-k loop of every trajectory
    by one iteration, and all candidate profiles of a round are evaluated in ONE likelihood call.

    ``driver='native'`` (what ``'auto'`` picks whenever it applies): the loops run inside the native inference driver
    (csrc/run_host.cpp, `_lib.RunHandle`) -- Python is touched once per round, for three bulk draws from the global NumPy
    stream (``standard_gamma``, ``random_sample``, ``standard_normal``), not once per sampler step.  Per trajectory the
    random numbers are consumed in the reference's order, so ``sample_many([traj], model)`` walks through the same random
    numbers as ``sample(traj, model)`` and gives the same result bit for bit; with several trajectories the stream is
    shared round by round (every run is a valid run of `sample`, none is the one a sequential call would have made).
    ``rng``: a ``numpy.random.Generator`` to draw the rounds' random numbers from instead of the global legacy stream (the
    reference's) -- the same three bulk draws per round through the Generator's ziggurat / modern gamma code, 2-3 times
    faster; with draws the largest part of a native run's wall time (BASELINE configs[4]: 0.12 of 0.20 s), that is most of
    what is left to gain on the host.  Native driver only.
    It applies to the default keywords of `sample` and to ``sampler_kw`` within {N (< 2000), concentration_brake,
    polarization_brake, max_fev, max_fcomplete}, ``choice_kw`` within {samplesize}, and to models that are a plain
    `MultiStateRouse` (GPU likelihood) or offer ``logL_segments(seg_start, seg_state, trajs, traj_id)``.
    ``driver='python'``: one unmodified `sample` loop per trajectory as cooperative tasks whose pending AMIS batches are
    fused into one launch (`batching.run_batched`; any keyword `sample` takes).

    Returns a list of `SamplingResults`, one per trajectory, in order.  With ``return_exceptions=True``
    a trajectory whose loop raised gets the exception object as its entry instead of aborting the rest.
    """
    if driver not in ('auto', 'native', 'python'):
        raise ValueError("driver must be 'auto', 'native' or 'python'")
    trajs = list(trajs)
    if driver != 'python':
        plan, why = _native_plan(model, kwargs)
        if plan is not None:
            return _sample_many_native(trajs, model, return_exceptions, plan, rng)
        if driver == 'native':
            raise ValueError("the native inference driver does not apply: " + why)
    if rng is not None:
        raise ValueError("rng= is served by the native inference driver only" + ("" if driver == 'python' else ": " + why))
    from .batching import run_batched
    results = run_batched(trajs, model, sample, return_exceptions=return_exceptions, **kwargs)
    for res in results:     # the loops saw a batching proxy of the model: hand the real one back
        if isinstance(res, SamplingResults):
            res.model = model
            for sampler in res.samplers:
                sampler.model = model
    return results


_SAMPLE_DEFAULTS = dict(dE=0, init_runs=20, certainty_in_k=0.99, k_lookahead=2, k_max=20, sampler_kw={}, choice_kw={}, show_progress=False)
_SAMPLER_DEFAULTS = dict(N=100, concentration_brake=1e-2, polarization_brake=1e-3, max_fev=20000, max_fcomplete=1000)
_per_k_cache = {}


def _native_plan(model, kwargs):
    """ (settings for the native driver, None) when it can run this call, else (None, the reason) """
    from .models import MultiStateRouse
    unknown = set(kwargs) - set(_SAMPLE_DEFAULTS)
    if unknown:
        return None, f"unknown keywords {sorted(unknown)}"
    kw = dict(_SAMPLE_DEFAULTS, **kwargs)
    sampler_kw, choice_kw = dict(kw['sampler_kw']), dict(kw['choice_kw'])
    if kw['show_progress']:
        return None, "show_progress"
    if set(sampler_kw) - set(_SAMPLER_DEFAULTS):
        return None, f"sampler_kw {sorted(set(sampler_kw) - set(_SAMPLER_DEFAULTS))}"
    if set(choice_kw) - {'samplesize'}:
        return None, f"choice_kw {sorted(set(choice_kw) - {'samplesize'})}"
    smp = dict(_SAMPLER_DEFAULTS, **sampler_kw)
    if smp['N'] >= 2000:
        return None, "N >= 2000 per step (the fused device step of `FixedkSampler` serves such batches)"
    if type(model) is MultiStateRouse:
        mode = 'gpu'
    elif hasattr(model, 'logL_segments'):
        mode = 'segments'
    else:
        return None, "the model offers no batch entry over several trajectories"
    if not all(float(v).is_integer() and v >= 0 for v in (kw['init_runs'], kw['k_lookahead'], kw['k_max'], smp['N'], smp['max_fev'],
                                                            smp['max_fcomplete'], choice_kw.get('samplesize', 10000))) or smp['N'] < 1:
        return None, "non-integer settings"
    return dict(mode=mode, dE=float(kw['dE']), init_runs=int(kw['init_runs']), certainty_in_k=float(kw['certainty_in_k']),
                k_lookahead=int(kw['k_lookahead']), k_max=int(kw['k_max']), samplesize=int(choice_kw.get('samplesize', 10000)),
                sampler=dict(N=int(smp['N']), concentration_brake=float(smp['concentration_brake']),
                             polarization_brake=float(smp['polarization_brake']), max_fev=int(smp['max_fev']),
                             max_fcomplete=int(smp['max_fcomplete']))), None


def _per_k_constants(transitions, k_max, Nmax):
    """
    what `FixedkSampler.__init__` derives from (transitions, k) alone, for k = 0 .. k_max: the uniform CFC weights, the
    log-prior of a profile, the number of valid traces and -- where they may be enumerated -- the traces themselves
    (bild/amis.py:650-659, 776).  Cached per transition matrix: twenty milliseconds of exact integer path counting.
    """
    from .amis import CFC, Dirichlet
    transitions = np.asarray(transitions, dtype=bool)
    key = (transitions.shape[0], transitions.tobytes())
    hit = _per_k_cache.setdefault(key, {'cfc': CFC(transitions), 'dirichlet': Dirichlet(), 'k': {}})
    cfc, out = hit['cfc'], []
    for k in range(k_max + 1):
        if k not in hit['k']:
            n_total = cfc.N_total(k)
            hit['k'][k] = (cfc.logp_uniform(k), float(np.sum(np.log(np.arange(k) + 1))) - cfc.N_total(k, log=True), n_total, {})
        logp0, logprior, n_total, traces = hit['k'][k]
        tr = None
        if n_total <= Nmax:
            if 'all' not in traces:
                traces['all'] = cfc.full_sample(k, Nmax=n_total)
            tr = traces['all']
        out.append((logp0, logprior, float(min(n_total, 10 ** 300)), tr))
    return hit, out


def _sample_many_native(trajs, model, return_exceptions, plan, rng=None, interruptible=False):
    """ `sample_many` through the native inference driver (see there); ``interruptible``: a KeyboardInterrupt between two
        rounds ends the run with what has been sampled so far, as `sample` does """
    from . import _lib
    from .amis import FixedkSampler
    trajs = [make_trajectory(t) for t in trajs]
    if not trajs:
        return []
    smp = plan['sampler']
    shared, per_k = _per_k_constants(model.transitions, plan['k_max'], min(smp['max_fcomplete'], smp['max_fev']))
    settings = dict(init_runs=plan['init_runs'], k_lookahead=plan['k_lookahead'], k_max=plan['k_max'], reserved=0,
                    certainty_in_k=plan['certainty_in_k'], dE=plan['dE'], N=smp['N'], concentration_brake=smp['concentration_brake'],
                    polarization_brake=smp['polarization_brake'], max_fev=smp['max_fev'], max_fcomplete=smp['max_fcomplete'],
                    choice_samplesize=plan['samplesize'])
    lengths = [len(t) for t in trajs]
    run = _lib.RunHandle(lengths, model.transitions, settings, per_k)
    gpu = plan['mode'] == 'gpu'
    if gpu:
        handle, ts = model.handle(), model.trajset(trajs)
    else:
        T_arr, n_states = np.asarray(lengths, dtype=np.int32), int(model.nStates)
    def raise_first_failure():      # a loop that raised ends the call, as an exception out of `sample` would
        for j in range(len(trajs)):
            info = run.traj_info(j)
            if info[0] == 2:
                raise _native_error(info)

    if rng is None:     # the global legacy stream: what the reference draws from
        draw_gamma, draw_uniform, draw_normal = np.random.standard_gamma, np.random.random_sample, np.random.standard_normal
    else:
        draw_gamma, draw_uniform, draw_normal = rng.standard_gamma, rng.random, rng.standard_normal
    failed_before = 0
    try:
        while True:
            counts, shapes = run.plan()
            n_gamma, n_uniform, n_normal, n_rows, live = (int(v) for v in counts[:5])
            if not return_exceptions and int(counts[6]) > failed_before:
                raise_first_failure()
            if n_rows == 0 and live == 0:
                break
            # the round's random numbers: three bulk draws from the global NumPy stream
            gammas = draw_gamma(shapes) if n_gamma else np.empty(0)
            uniforms = draw_uniform(n_uniform)
            normals = draw_normal(n_normal)
            if gpu:
                run.round(handle, ts, gammas, uniforms, normals, path=model.path)
            else:
                ss, thetas, tid = run.stage(gammas, uniforms)
                if len(tid):
                    seg_start, seg_state = _lib.segments_from_st(ss, thetas, T_arr[tid], n_states)
                    logLs = np.asarray(model.logL_segments(seg_start, seg_state, trajs, tid), dtype=np.float64)
                else:
                    logLs = np.empty(0)
                run.finish(logLs, normals)
    except KeyboardInterrupt:  # pragma: no cover
        if not interruptible:
            raise
    results = []
    for j, traj in enumerate(trajs):
        state, kind, n_samplers, n_rows, width, message = run.traj_info(j)
        if state == 2:
            results.append(_native_error((state, kind, n_samplers, n_rows, width, message)))
            continue
        samplers = []
        for k in range(n_samplers):
            info = run.sampler_info(j, k)
            data = run.sampler_data(j, k, info[3], info[4])
            core = run.take_core(j, k) if info[0] == 2 else None
            samplers.append(FixedkSampler._adopt_native(traj, model, k, smp, (shared['dirichlet'], shared['cfc'], per_k[k][1]) if k < len(per_k) else None,
                                                        info, data, core))
        ks, flags, i_la, pk, kld = run.traj_log(j, n_rows, width)
        flags = flags.tolist()
        log = {'k': ks.tolist(),
               'pk': [pk[i, :(f >> 8) & 255] if f & 1 else None for i, f in enumerate(flags)],
               'KLD': [kld[i, :(f >> 16) & 255] if f & 2 else None for i, f in enumerate(flags)],
               'I_la': [float(i_la[i]) if f & 4 else None for i, f in enumerate(flags)]}
        results.append(SamplingResults(traj, model, plan['dE'], samplers, log))
    return results


def _native_error(info):
    kind, message = info[1], info[5]
    if kind == 1:
        return RuntimeError(message)
    if kind == 2:
        return ValueError(message)
    from . import _lib
    return _lib.BildAmdError(1, message)


class _NoBar:
    def update(self):
        pass

    def close(self):
        pass


def _progress_bar(show):
    if not show:
        return _NoBar()
    try:
        from tqdm.auto import tqdm
        return tqdm()
    except ImportError:  # pragma: no cover
        return _NoBar()


class SamplingResults:
    """
    Output of `sample` (reference bild/core.py:238-372).

    Attributes: ``traj``, ``model``, ``dE``, ``samplers`` (list of FixedkSampler, index = k),
    ``log`` (dict of nan-padded arrays: 'k', 'I_la' 1-d; 'pk', 'KLD' 2-d), and the properties
    ``k``, ``evidence``, ``evidence_se``.
    """

    def __init__(self, traj, model, dE, samplers, log=None):
        self.traj = traj
        self.model = model
        self.dE = dE
        self.samplers = samplers
        self.log = {}
        if log is not None:
            for key, rows in log.items():
                if key in ('k', 'I_la'):
                    self.log[key] = np.array(rows)
                else:
                    width = max([1] + [len(r) for r in rows if r is not None])
                    arr = np.full((len(rows), width), np.nan)
                    for i, r in enumerate(rows):
                        if r is not None:
                            arr[i, :len(r)] = r
                    self.log[key] = arr

    @property
    def k(self):
        return np.array([s.k for s in self.samplers])

    @property
    def evidence(self):
        return np.array([s.evidences[-1][0] for s in self.samplers])

    @property
    def evidence_se(self):
        return np.array([s.evidences[-1][1] for s in self.samplers])

    def best_k(self, dE=None):
        """ smallest k whose evidence is within dE of the maximum (bild/core.py:304-326) """
        if dE is None:
            dE = self.dE
        return np.min(self.k[self.evidence >= np.max(self.evidence) - dE])

    def best_profile(self, dE=None):
        """ MAP profile of the sampler at `best_k` (bild/core.py:328-346) """
        return self.samplers[self.best_k(dE)].MAP_profile()

    def log_marginal_posterior(self, dE=None):
        """
        (n, T) log posterior state marginals (bild/core.py:348-372); ``dE='average'`` averages
        over k weighted by evidence instead of picking the best k.
        """
        if isinstance(dE, str) and dE == 'average':
            with np.errstate(under='ignore'):
                logpost = logsumexp([s.log_marginal_posterior() + logev
                                     for s, logev in zip(self.samplers, self.evidence)
                                     if s.evidences[-1][0] > -np.inf], axis=0)
                return logpost - logsumexp(logpost, axis=0)
        if dE is None:
            dE = self.dE
        return self.samplers[self.best_k(dE)].log_marginal_posterior()
