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
    traj = make_trajectory(traj)
    progress = _progress_bar(show_progress)

    samplers = []
    log = {'k': [], 'pk': [], 'KLD': [], 'I_la': []}
    state = {'fresh': False}

    def add_sample(k):
        if samplers[k].step():   # an exhausted sampler does nothing
            progress.update()
            for key in log:
                log[key].append(None)
            log['k'][-1] = k
            state['fresh'] = True

    def add_sampler(k):
        assert k == len(samplers)
        samplers.append(FixedkSampler(traj, model, k=k, **sampler_kw))
        for _ in range(init_runs):
            add_sample(k)

    def next_k():
        k_new = len(samplers)
        if not state['fresh']:
            return k_new if len(log['k']) == 0 else log['k'][-1]

        logE = np.array([s.evidences[-1][0] for s in samplers])
        dlogE = np.array([s.evidences[-1][1] for s in samplers])
        nsteps = np.array([np.inf if s.exhausted else len(s.samples) for s in samplers])
        cs = ChoiceSampler(logE, dlogE ** 2, nsteps, dE, **choice_kw)
        pk = cs.n0 / cs.samplesize

        # look-ahead region = the last k_lookahead samplers.  While every sampler is still inside
        # it, its importance is infinite: open the next k right away (if allowed).
        if k_new < k_lookahead + 1 and k_new <= k_max:
            choice, KLD, I_la = k_new, None, np.inf
        else:
            KLD = cs.KLD_moreSamples()
            choice = int(np.argmax(KLD))
            I_la = cs.KLD_omitK(np.arange(k_new - k_lookahead, k_new)) if k_new >= k_lookahead + 1 else np.inf
            if I_la > KLD[choice] and k_new <= k_max:
                choice = k_new

        log['pk'][-1] = pk
        log['KLD'][-1] = KLD
        log['I_la'][-1] = I_la
        state['fresh'] = False
        return choice

    k_next = 0
    running = True
    try:
        while running:
            if k_next < len(samplers):
                add_sample(k_next)
            elif k_next == len(samplers):
                add_sampler(k_next)
            else:  # pragma: no cover
                raise RuntimeError("Trying to sample outside of existing range; this is a bug")

            progressed = state['fresh']
            k_next = next_k()

            if k_next == len(samplers):
                # a new k takes precedence over the certainty criterion.  (When no sampler has taken a single step
                # yet -- every k so far was enumerated exhaustively or has k >= T, i.e. a trajectory of a few frames
                # -- the reference keeps opening samplers without looking at k_max, forever; here k_max ends it.)
                running = k_next <= k_max
            elif not progressed and samplers[k_next].exhausted:
                # nothing was sampled in this round and the next candidate cannot sample either: the reference
                # would spin on it; there is nothing left to learn
                running = False
            else:
                running = np.max(log['pk'][-1]) < certainty_in_k
                if log['KLD'][-1] is not None:
                    # no information left to gain when all relevant samplers are exhausted
                    running = running and log['KLD'][-1][k_next] > 0
        progress.close()
    except KeyboardInterrupt:  # pragma: no cover
        pass
    return SamplingResults(traj, model, dE, samplers, log)


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
