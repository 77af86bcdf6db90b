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
           ):
    """
    Run BILD on one trajectory (reference bild/core.py:22-236): AMIS samplers for k = 0, 1, ...
    are created on demand; after each step the choice distribution p(k) decides whether to
    refine an existing k, open the next one (look-ahead rule), or stop
    (``max p(k) >= certainty_in_k``).

    Parameters and defaults are those of the reference.  Returns `SamplingResults` -- also on
    ``KeyboardInterrupt``, with whatever has been sampled so far.
    """
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


def sample_many(trajs, model, **kwargs):
    """
    `sample` for a list of trajectories, run concurrently; the AMIS batches that are pending at
    the same time are evaluated together in one launch per round (see `batching`).

    Returns a list of `SamplingResults`, one per trajectory, in order.  With ``return_exceptions=True``
    a trajectory whose loop raised gets the exception object as its entry instead of aborting the rest.
    """
    from .batching import run_batched
    results = run_batched(trajs, model, sample, **kwargs)
    for res in results:     # the loops saw a batching proxy of the model: hand the real one back
        if isinstance(res, SamplingResults):
            res.model = model
            for sampler in res.samplers:
                sampler.model = model
    return results


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
