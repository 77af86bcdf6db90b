"""
Which k should be sampled next?  Information-based sample selection for the adaptive-k loop.

Counterpart of reference bild/choicesampler.py:3-210.  Host-side
Monte Carlo on <= samplesize x k_max floats; nothing here touches the GPU.

The *choice distribution* p(k) is the belief about which k is best under the evidence margin
dE, obtained by sampling the evidence curve within its error bars; the value of an extra AMIS
step at k is the expected Kullback-Leibler divergence between the updated and the current
choice distribution.
"""
import numpy as np


class ChoiceSampler:
    """
    Parameters (reference bild/choicesampler.py:83-99)
    ----------
    muhat : (k,) point estimates of the log-evidence
    shat : (k,) their variances (squared standard errors)
    N : (k,) number of AMIS steps behind each estimate (np.inf for exhausted samplers)
    dE : evidence margin
    samplesize : size of the underlying standard-normal sample
    """

    def __init__(self, muhat, shat, N, dE, samplesize=10000):
        self.dE = dE
        self.muhat = muhat
        self.shat = shat
        self.N = N
        self.samplesize = samplesize
        self.kmax = len(muhat)
        # expected squared shift of the estimate at k after one more step, and its root
        self.EDmu2 = self.shat / (self.N + 1)
        self.Dmu = np.sqrt(self.EDmu2)
        self.init_sample()

    def init_sample(self):
        """ (re)draw the common random numbers all evaluations share (bild/choicesampler.py:101-113) """
        self._scaled_rvs = np.sqrt(self.shat[None, ...]) * np.random.normal(size=(self.samplesize, self.kmax))
        self.bestk = self.evaluate()
        self.best_is_k = self.bestk[:, None] == np.arange(self.kmax)[None, :]
        self.n0 = np.sum(self.best_is_k, axis=0)

    def evaluate(self, k_change=None, n_step=0, omit_k=None):
        """
        One draw of "the best k" per row of the common sample (bild/choicesampler.py:115-153):
        the smallest k whose (perturbed) evidence is within dE of the row maximum.  ``k_change``
        shifts the estimate at those k by ``n_step`` natural step sizes; ``omit_k`` ignores k's.
        """
        mu = self.muhat.copy()
        if k_change is not None:
            mu[k_change] += n_step * self.Dmu[k_change]
        if omit_k is not None:
            mu[omit_k] = np.nan
        x = self._scaled_rvs + mu
        top = np.nanmax(x, axis=1, keepdims=True)
        return np.nanargmax(top - self.dE - x <= 0, axis=1)

    def Dn(self):
        """
        (k_change, k) expected change of the choice histogram per extra step (bild/choicesampler.py:155-169):
        2 k_max evaluations of the whole sample in the reference, one pass of native host code here
        (`Dn_numpy` is the same thing in array operations; tests compare them).
        """
        from . import _lib
        return _lib.choice_counts(self._scaled_rvs, self.muhat, self.Dmu, self.dE)[1]

    def Dn_numpy(self):
        counts = []
        for step in (-0.5, 0.5):
            ks = np.array([self.evaluate(k, step) for k in range(self.kmax)])       # (k_change, samp)
            counts.append(np.sum(ks[..., None] == np.arange(self.kmax), axis=-2))   # (k_change, k)
        return counts[1] - counts[0]

    def KLD_moreSamples(self):
        """ (k,) expected information gain of one more step at each k (bild/choicesampler.py:171-181) """
        Dn = self.Dn()
        return 0.5 / self.samplesize * np.sum(Dn ** 2 / (self.n0 + 1)[None, :], axis=-1)

    def KLD_omitK(self, omit_k=None):
        """
        Information carried by the positions ``omit_k``: KL(full || omitted)
        (bild/choicesampler.py:183-210).
        """
        from . import _lib
        n_without = _lib.choice_counts(self._scaled_rvs, self.muhat, self.Dmu, self.dE, omit=omit_k, want_dn=False)[2]
        n_without = n_without / np.sum(n_without) * self.samplesize
        Dn = self.n0 - n_without
        Dn[omit_k] = 0  # the omitted slots themselves would contribute an infinite divergence
        return 0.5 / self.samplesize * np.sum(Dn ** 2 / (n_without + 1))
