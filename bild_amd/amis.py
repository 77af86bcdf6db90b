"""
AMIS posterior sampling at a fixed number of switches k.

Counterpart of reference bild/amis.py: the proposal families
(`Dirichlet` over switch intervals, `CFC` over state traces) and the `FixedkSampler` whose
``logL(ss, thetas)`` is the batch boundary of the GPU likelihood (bild/amis.py:717-739).

Everything here except ``FixedkSampler.logL`` is host-side O(N k) bookkeeping.  Behaviour follows the
reference function by function (citations in the docstrings); the random streams are consumed in the
same order (``np.random.dirichlet`` as ``scipy.stats.dirichlet.rvs`` calls it, ``np.random.choice``,
``np.random.rand``), so that with the same ``np.random.seed`` and the same likelihood values a sampler
of this module and a reference sampler walk through identical proposals, weights and evidences
(tests/test_amis.py, against runs of the reference stored in tests/golden/amis_*.npz).  The per-step
bookkeeping of `FixedkSampler` exists twice: in NumPy here (the specification) and as native host code
behind the C ABI (csrc/amis_host.cpp, the default).

Profiles are parametrised as ``(s, theta)``: ``s`` (k+1,) interval lengths on the unit
simplex, ``theta`` (k+1,) the state of each interval; neighbouring states obey the model's
``transitions`` matrix.
"""
import itertools
import os
import math

import numpy as np
from scipy.special import gammaln, xlogy

from .profiles import Loopingprofile, segments_from_st, switch_indices


# ----------------------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------------------
def logsumexp(a, axis=None, keepdims=False, mask=None):
    """
    log(sum(exp(a))) over ``axis`` (over the entries selected by ``mask``, if given), shifted by the largest
    entry; an empty or all ``-inf`` selection gives ``-inf``.  (scipy.special.logsumexp computes the same;
    its argument handling costs more than the arithmetic on the small arrays of an AMIS step.)
    """
    a = np.asarray(a, dtype=np.float64)
    if mask is not None:
        a = np.where(mask, a, -np.inf)
    top = np.max(a, axis=axis, keepdims=True)
    top = np.where(np.isfinite(top), top, 0.)
    with np.errstate(under='ignore', divide='ignore'):
        out = np.log(np.sum(np.exp(a - top), axis=axis, keepdims=True)) + top
    if keepdims:
        return out
    return out.reshape(())[()] if axis is None else np.squeeze(out, axis=axis)


def _masked_logsumexp(logx, mask, axis):
    """ log(sum(exp(logx) * mask)) without warnings for empty selections (-> -inf) """
    return logsumexp(logx, axis=axis, mask=mask)


def _int_matrix_power(T, p):
    """ exact integer matrix power (python ints: path counts overflow int64 for long traces) """
    n = len(T)
    result = [[int(i == j) for j in range(n)] for i in range(n)]
    base = [[int(v) for v in row] for row in T]

    def mul(A, B):
        return [[sum(A[i][l] * B[l][j] for l in range(n)) for j in range(n)] for i in range(n)]
    while p > 0:
        if p & 1:
            result = mul(result, base)
        base = mul(base, base)
        p >>= 1
    return result


def _safe_log(x):
    return math.log(x) if x > 0 else -np.inf


# ----------------------------------------------------------------------------------------
# proposal over switch intervals
# ----------------------------------------------------------------------------------------
class Dirichlet:
    """ Dirichlet distribution with a weighted method-of-moments fit (bild/amis.py:59-151) """

    def sample(self, a, N=1):
        """ (N, k+1) draws (bild/amis.py:66-81) """
        a = np.asarray(a, dtype=np.float64)
        ss = np.random.dirichlet(a, size=N)   # the call scipy.stats.dirichlet.rvs makes
        if np.isfinite(ss.sum()):             # (a NaN or Inf anywhere shows in the sum: one pass instead of three)
            return ss
        bad = ~np.all(np.isfinite(ss), axis=1)
        if np.any(bad):
            # All concentrations tiny (a bimodal posterior at the corners of the simplex drives their sum towards
            # 0): every gamma variate of a draw underflows and NumPy returns 0/0.  The distribution is then, to all
            # digits, a mixture of point masses at the corners with probabilities a_i / sum(a): draw from that.
            # (Extra random numbers are consumed only here, where the reference goes on with NaN samples.)
            corners = np.random.choice(len(a), size=int(np.sum(bad)), p=a / np.sum(a))
            ss[bad] = np.eye(len(a))[corners]
        return ss

    def logpdf(self, a, ss, log_ss=None):
        """
        log density at the rows of ``ss`` (bild/amis.py:83-108).

        A sample with ``s_i == 0`` where ``a_i < 1`` sits on a pole of the density: ``+inf``
        (pinned by reference tests/test_amis.py:51-54).

        ``log_ss`` (see `log_samples`) lets a caller that evaluates many parameter vectors on the same
        samples pay for the logarithms once: the sum over components becomes a matrix-vector product.
        """
        return self.logpdf_many(np.asarray(a, dtype=np.float64)[None, :], ss, log_ss)[0]

    @staticmethod
    def log_samples(ss):
        """
        (log(ss) transposed to (k+1, N) with zeros where ss == 0, indices of the rows of ``ss`` that contain
        a zero): what `logpdf(_many)` needs of the samples, for callers that evaluate them repeatedly
        """
        ss = np.atleast_2d(np.asarray(ss, dtype=np.float64))
        zero = ss == 0
        with np.errstate(divide='ignore'):
            logs = np.log(ss.T)     # a copy in (k+1, N) order: sums over samples run along contiguous rows
        if np.any(zero):
            logs[zero.T] = 0.
        return logs, np.nonzero(np.any(zero, axis=1))[0]

    def logpdf_many(self, As, ss, log_ss=None):
        """ log densities of the rows of ``ss`` under each concentration vector of ``As`` (P, k+1) -> (P, N) """
        As = np.atleast_2d(np.asarray(As, dtype=np.float64))
        log_norm = gammaln(np.sum(As, axis=1)) - np.sum(gammaln(As), axis=1)            # (P,)
        logs, zero_rows = self.log_samples(ss) if log_ss is None else log_ss
        # einsum, not BLAS: single-threaded and summed in a fixed order (runs are reproducible bit for bit)
        out = np.einsum('pj,jn->pn', As - 1., logs) + log_norm[:, None]
        if len(zero_rows):   # x log(0): 0 for x = 0, -+inf else; poles of the density are +inf
            sz = np.atleast_2d(np.asarray(ss, dtype=np.float64))[zero_rows]
            with np.errstate(divide='ignore', invalid='ignore'):
                oz = log_norm[:, None] + np.sum(xlogy(As[:, None, :] - 1., sz[None, :, :]), axis=2)
            oz[np.any((sz[None, :, :] == 0) & (As[:, None, :] < 1), axis=2)] = np.inf
            out[:, zero_rows] = oz
        return out

    def estimate(self, ss, log_weights, ssT=None):
        """
        Weighted method of moments (bild/amis.py:110-151): mean m, variance v per component,
        total concentration ``A = mean(m (1-m) / v) - 1``, estimate ``A m``.  ``ssT``: the samples as a
        contiguous (k+1, N) array, if the caller keeps one.

        Weights below 1e-100 of the largest are taken as zero: they cannot change a double-precision sum
        that the largest weight dominates, but their products with small deviations are subnormal, and
        arithmetic on subnormals is slow enough to dominate an AMIS step over 1e5 pooled samples.
        """
        if ssT is None:
            ssT = np.ascontiguousarray(np.asarray(ss, dtype=np.float64).T)
        with np.errstate(under='ignore'):
            w = np.exp(log_weights - np.max(log_weights))
        w[w < 1e-100] = 0.
        w = w / np.sum(w)
        m = np.empty(len(ssT))
        v = np.empty(len(ssT))
        for j, col in enumerate(ssT):    # plain reductions over contiguous rows (no BLAS: fixed summation order)
            m[j] = np.sum(col * w)
            dev = col - m[j]
            dev *= dev
            dev *= w
            v[j] = np.sum(dev)
        if np.any(v == 0):
            total = 1e10  # degenerate sample: very concentrated but finite, the brake takes over
        else:
            total = np.mean(m * (1 - m) / v) - 1
        # Weight on the boundary of the simplex (a proposal that has collapsed there: m_i = 1 up to rounding, or
        # m_i = 0) makes the formula return a negative / zero / non-finite concentration, with which the reference
        # ends in ValueError("alpha <= 0").  Treated like the degenerate case above; components stay positive.
        if not (np.isfinite(total) and total > 0):
            total = 1e10
        return np.maximum(total * m, np.finfo(np.float64).tiny)


# ----------------------------------------------------------------------------------------
# proposal over state traces
# ----------------------------------------------------------------------------------------
class CFC:
    """
    Conflict Free Categorical (bild/amis.py:153-536): a distribution over state traces
    ``theta`` (k+1 integers in [0, n)) whose consecutive entries must be allowed by
    ``transitions``.  Parametrised by log-weights ``logp`` (n, k+1); sampling is causal: slot
    ``i`` is drawn from ``p[:, i]`` restricted to the successors of ``theta[i-1]``.
    """

    def __init__(self, transitions):
        self.transitions = np.array(transitions, dtype=bool, copy=True)
        self.MOM_maxiter = 1000
        self.MOM_precision = 1e-2

    @property
    def n(self):
        return self.transitions.shape[0]

    # -- sampling / evaluation ---------------------------------------------------------
    def sample(self, logp, N=1):
        """ (N, k+1) traces (bild/amis.py:223-256); same use of the global NumPy stream """
        k1 = logp.shape[1]
        assert k1 >= 1
        with np.errstate(under='ignore'):
            p = np.exp(logp - logsumexp(logp, axis=0))
        thetas = np.empty((N, k1), dtype=int)
        thetas[:, 0] = np.random.choice(self.n, size=N, p=p[:, 0])
        for i in range(1, k1):
            allowed = p[None, :, i] * self.transitions[thetas[:, i - 1]]  # (N, n)
            cdf = np.cumsum(allowed, axis=1)
            cdf /= cdf[:, [-1]]
            thetas[:, i] = np.argmax(cdf > np.random.rand(N, 1), axis=1)  # first crossing
        return thetas

    def logpmf(self, logp, thetas, codes=None):
        """
        log probability of each trace (bild/amis.py:258-282).

        The probability of slot i >= 1 depends only on (i, theta[i-1], theta[i]): the n*n*k possible
        terms are tabulated once (`lookup_tables`) and gathered through per-sample integer codes
        (`trace_codes`), instead of one masked logsumexp per sample as in the reference; the values are
        the same.  (With the GPU likelihood this bookkeeping, re-run for every earlier sample at every
        AMIS step, is what an AMIS step costs.)
        """
        return self.logpmf_many(np.asarray(logp, dtype=float)[None], thetas, codes)[0]

    def trace_codes(self, thetas):
        """ (theta[0], flat indices (N, k) into the pair table of `lookup_tables`) of the traces """
        thetas = np.asarray(thetas)
        n = self.n
        k = thetas.shape[1] - 1
        pair = (np.arange(k)[None, :] * n + thetas[:, :-1]) * n + thetas[:, 1:]
        return thetas[:, 0].astype(np.intp), pair.astype(np.intp)

    def lookup_tables(self, logps):
        """
        (P, n) log probabilities of the first state and (P, k*n*n) of every (slot, previous, current)
        triple under each weight matrix of ``logps`` (P, n, k+1).

        A state of weight exactly zero has probability zero -- also when every successor allowed after
        its predecessor has weight zero, where the reference's ``-inf - (-inf)`` gives NaN (possible with
        a 2-state model once the weights of a very peaked posterior have underflowed; the NaN then
        poisons every weight and ends in "Iteration did not converge").
        """
        logps = np.asarray(logps, dtype=float)
        P, n, k1 = logps.shape
        with np.errstate(under='ignore'):
            head = logps[:, :, 0] - logsumexp(logps[:, :, 0], axis=1, keepdims=True)       # (P, n)
        # norm[p, i-1, prev] = logsumexp of logps[p, :, i] over the successors of state prev
        by_slot = np.swapaxes(logps, 1, 2)[:, 1:, :]                                        # (P, k, n) current state last
        norm = _masked_logsumexp(by_slot[:, :, None, :], self.transitions[None, None, :, :], axis=-1)   # (P, k, prev)
        with np.errstate(invalid='ignore'):
            pair = by_slot[:, :, None, :] - norm[:, :, :, None]                             # (P, k, prev, cur)
        pair = np.where(by_slot[:, :, None, :] == -np.inf, -np.inf, pair)
        return head, pair.reshape(P, -1)

    def logpmf_many(self, logps, thetas, codes=None, tables=None):
        """
        log probabilities of the traces under each weight matrix of ``logps`` (P, n, k+1) -> (P, N);
        ``codes`` = `trace_codes` of the traces and ``tables`` = `lookup_tables` of the matrices, if the
        caller keeps them.
        """
        first, pair = self.trace_codes(thetas) if codes is None else codes
        head, table = self.lookup_tables(logps) if tables is None else tables
        out = head[:, first]
        if pair.shape[1]:
            out = out + np.sum(table[:, pair], axis=-1)
        return out

    # -- estimation -------------------------------------------------------------------
    def estimate(self, thetas, log_weights):
        """
        weighted marginals per slot -> weight parameters (bild/amis.py:284-307).  The marginals are
        accumulated with one weighted histogram per slot (shifted by the largest log-weight) instead of
        a masked logsumexp over an (n, N, k+1) indicator array.
        """
        thetas = np.asarray(thetas)
        log_weights = np.asarray(log_weights, dtype=float)
        top = np.max(log_weights)
        with np.errstate(under='ignore'):
            w = np.exp(log_weights - top) if np.isfinite(top) else np.zeros_like(log_weights)
        marg = np.stack([np.bincount(thetas[:, i], weights=w, minlength=self.n)[:self.n]
                         for i in range(thetas.shape[1])], axis=1)                      # (n, k+1)
        with np.errstate(divide='ignore', under='ignore'):
            log_marginals = np.log(marg) + top
            log_marginals = log_marginals - logsumexp(log_marginals, axis=0, keepdims=True)
        return self.logp_from_marginals(log_marginals)

    def logp_from_marginals(self, log_marginals):
        """ slot-by-slot inversion marginals -> weights (bild/amis.py:309-337) """
        k1 = log_marginals.shape[1]
        assert k1 >= 1
        logp = np.empty(log_marginals.shape, dtype=float)
        logp[:, 0] = log_marginals[:, 0]
        for i in range(1, k1):
            logp[:, i] = self.solve_marginals_single(log_marginals[:, i], log_marginals[:, i - 1])
        return logp

    def solve_marginals_single(self, logf, logg):
        """
        Fixed-point iteration  p_n = f_n / sum_{m -> n} g_m / (sum_{m -> j} p_j)
        (bild/amis.py:339-399).  Stops when successive iterates differ by less than
        ``MOM_precision`` in log space; ``RuntimeError`` after ``MOM_maxiter`` iterations.
        """
        logf = np.asarray(logf, dtype=float)
        logg = np.asarray(logg, dtype=float)
        # delta-like marginals need no iteration
        if np.any(logf == 0):
            return logf.copy()
        if np.any(logg == 0):
            assert np.all(logf[logg == 0] == -np.inf)
            return logf.copy()

        f_zero = logf == -np.inf
        g_zero = logg == -np.inf
        cur = logf
        for _ in range(self.MOM_maxiter):
            with np.errstate(under='ignore', invalid='ignore'):
                out_norm = _masked_logsumexp(cur[None, :], self.transitions, axis=1)   # successors of m
                out_norm[g_zero] = 0                                                    # avoid -inf + inf
                flow = logg - out_norm
                inflow = _masked_logsumexp(flow[:, None], self.transitions, axis=0)    # predecessors of n
                inflow[f_zero] = 0
                new = logf - inflow
                new = new - logsumexp(new)
            if np.max(np.abs(new[~f_zero] - cur[~f_zero])) < self.MOM_precision:
                return new
            cur = new
        raise RuntimeError("Iteration did not converge")

    # -- the uniform distribution over traces -----------------------------------------
    def _path_counts(self, k):
        """ powers T^0 .. T^k of the transition matrix with exact integers """
        T = self.transitions.astype(int).tolist()
        return [_int_matrix_power(T, i) for i in range(k + 1)]

    def uniform_marginals(self, k):
        """
        Slot marginals of the uniform distribution over valid traces (bild/amis.py:401-453):
        (#paths of length i ending in state) x (#paths of length k-i starting there).
        """
        n = self.n
        pw = self._path_counts(k)
        out = np.empty((n, k + 1), dtype=float)
        for i in range(k + 1):
            into = [sum(pw[i][a][s] for a in range(n)) for s in range(n)]          # column sums
            outof = [sum(pw[k - i][s][b] for b in range(n)) for s in range(n)]     # row sums
            counts = [into[s] * outof[s] for s in range(n)]
            total = sum(counts)
            out[:, i] = [_safe_log(c) - _safe_log(total) for c in counts]
        return out

    def logp_uniform(self, k):
        """ weight parameters of the uniform distribution (bild/amis.py:455-476) """
        return self.logp_from_marginals(self.uniform_marginals(k))

    def N_total(self, k, log=False):
        """ number of valid traces with k switches (bild/amis.py:478-497) """
        N = sum(sum(row) for row in _int_matrix_power(self.transitions.astype(int).tolist(), k))
        return math.log(N) if log else N

    def full_sample(self, k, Nmax=1000):
        """
        All valid traces with k switches, lexicographically by depth-first expansion
        (bild/amis.py:499-536); ``ValueError`` if there are more than ``Nmax``.
        """
        N = self.N_total(k)
        if N > Nmax:
            raise ValueError(f"Full sample would be {N} > Nmax = {Nmax} traces")
        succ = [np.nonzero(row)[0].tolist() for row in self.transitions]
        traces = [[s] for s in range(self.n)]
        for _ in range(k):
            traces = [tr + [nxt] for tr in traces for nxt in succ[tr[-1]]]
        return np.array(traces, dtype=int).reshape(len(traces), k + 1)


# ----------------------------------------------------------------------------------------
# the sampler
# ----------------------------------------------------------------------------------------
class _SampleList:
    """
    ``FixedkSampler.samples`` as the reference presents it -- one dict per AMIS step with the arrays
    'ss', 'thetas', 'logLs' [, 'logδs', 'log_weights', 'cur_log_proposal'] -- as views of the sampler's
    pooled arrays, materialised on access (a step itself never walks the list).
    """

    def __init__(self, sampler):
        self._s = sampler

    def __len__(self):
        return len(self._s._sizes)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("sample index out of range")
        s = self._s
        lo = int(sum(s._sizes[:i]))
        hi = lo + s._sizes[i]
        out = {'ss': s._pool['ss'][lo:hi], 'thetas': s._pool['thetas'][lo:hi]}
        for key, arr in s._arr.items():
            out[key] = arr[lo:hi]
        return out

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class FixedkSampler:
    """
    AMIS (Cornuet et al. 2012) for a fixed number of switches ``k``; one `step` draws ``N``
    profiles from the current proposal, evaluates them, re-weights all samples drawn so far
    against the mixture of all proposals used, and refits the proposal.

    Constructor, attributes and methods follow reference bild/amis.py:540-972.  The likelihood
    of a batch is obtained with ONE call when the model offers ``logL_st_batch`` (the GPU
    model does), else per sample as in the reference.
    """

    class ExhaustionImpractical(ValueError):
        pass

    def __init__(self, traj, model, k,
                 N=100,
                 concentration_brake=1e-2,
                 polarization_brake=1e-3,
                 max_fev=20000,
                 max_fcomplete=1000,
                 native=True,
                 device_bookkeeping=None,
                 fused=True,
                 rng='numpy',
                 seed=None,
                 ):
        self.k = k
        self.N = N
        self.native = native
        # with the bookkeeping on the GPU: likelihood and bookkeeping of a step in one native call (`_step_native`)
        self.fused = fused
        # 'numpy' (default): the reference's random stream -- the global NumPy generator, consumed in the reference's order
        # (bild/amis.py:831-832).  'device': the samples of a step are drawn on the GPU from a counter-based generator keyed
        # by `seed` (csrc/amis_device.hip: draw_kernel) -- the same sampler in distribution, not the same random numbers;
        # applies where the fused step does (`_fusable`), else the NumPy stream is used.
        if rng not in ('numpy', 'device'):
            raise ValueError("rng must be 'numpy' or 'device'")
        self.rng = rng
        # (a default seed must not come out of the global NumPy stream: that stream is the reference's)
        self.seed = int.from_bytes(os.urandom(8), 'little') if seed is None else int(seed)
        self._device_drawn = 0      # pooled samples the device drew (the host fetches them when somebody looks)
        # where the native bookkeeping of a step runs: None = on the GPU for batches of >= 2000 samples per step when there
        # is one (the passes over the pooled samples are then most of a step), else on the host; True / False force it
        self.device_bookkeeping = device_bookkeeping
        self.brakes = (concentration_brake, polarization_brake)
        self.max_fev = max_fev
        self.max_fcomplete = max_fcomplete
        self.exhausted = False
        self.traj = traj
        self.model = model

        if self.k >= len(self.traj):
            # more switches than frames: unidentifiable by construction (bild/amis.py:641-648)
            self.evidences = [(-np.inf, 1e-10, np.inf)]
            self.exhausted = True
            return

        self.dirichlet = Dirichlet()
        self.cfc = CFC(model.transitions)
        self.parameters = [(np.ones(self.k + 1), self.cfc.logp_uniform(self.k))]
        # uniform prior over profiles: k! / N_total (bild/amis.py:654-659)
        self.logprior = float(np.sum(np.log(np.arange(self.k) + 1))) - self.cfc.N_total(self.k, log=True)

        # All samples drawn so far live in pooled arrays, in drawing order: `_pool` what is fixed at drawing
        # time ('ss', 'thetas' and derived lookup data), `_arr` what an AMIS step recomputes for every sample
        # ('logLs', 'logδs', 'cur_log_proposal', 'log_weights'); `_sizes` the number of samples per step.
        self._sizes = []
        self._pool_np = None
        self._chunks = []       # native path: (ss, thetas) per step, not yet appended to `_pool_np`
        self._arr_np = {}
        self._arr_cache = (None, None)
        self.samples = _SampleList(self)
        # The bookkeeping of `step` exists twice: in NumPy below (the specification, `native=False`) and as one
        # pass of native host code over the pooled samples (csrc/amis_host.cpp, the default), compared step by
        # step in tests/test_amis.py.
        self._core = None
        if native:
            self._core = self._new_core()
        # per proposal: concentration vector and CFC.lookup_tables, stacked
        self._As = np.empty((0, self.k + 1))
        self._heads = np.empty((0, self.cfc.n))
        self._tables = np.empty((0, self.k * self.cfc.n ** 2))
        self.evidences = []  # (logev, dlogev, KL) per step

        try:
            self.fix_exhaustive()
        except FixedkSampler.ExhaustionImpractical:
            pass

    @classmethod
    def _adopt_native(cls, traj, model, k, settings, shared, info, data, core):
        """
        A sampler whose whole history was produced by the inference driver (csrc/run_host.cpp, `core.sample_many`): the same
        object `__init__` + `step()` calls would have left behind, built without repeating any of their work.  The native
        bookkeeping (proposals, pooled samples and their arrays) is adopted as it is and fetched when somebody looks.

        settings : the sampler keywords of the run; shared : (Dirichlet, CFC, logprior) of this k;
        info : (kind, exhausted, steps, ...) from `RunHandle.sampler_info`; data : (evidences, ss, thetas, logLs);
        core : the adopted `_lib.AmisCore` or None.
        """
        self = object.__new__(cls)
        kind, exhausted, steps = info[0], info[1], info[2]
        evidences, ss, thetas, logLs = data
        self.k, self.N, self.native, self.fused, self.rng = k, settings['N'], True, True, 'numpy'
        self.seed = int.from_bytes(os.urandom(8), 'little')
        self._device_drawn, self._adopted, self.device_bookkeeping = 0, core is not None, None
        self.brakes = (settings['concentration_brake'], settings['polarization_brake'])
        self.max_fev, self.max_fcomplete = settings['max_fev'], settings['max_fcomplete']
        self.exhausted, self.traj, self.model = exhausted, traj, model
        self.evidences = [tuple(row) for row in evidences]
        if kind == 0:       # k >= T (bild/amis.py:641-648): nothing else exists
            return self
        self.dirichlet, self.cfc, self.logprior = shared
        self._sizes = [settings['N']] * steps if kind == 2 else [len(logLs)]
        self._pool_np = {'ss': ss, 'thetas': thetas} if kind == 1 else None
        self._chunks = []
        self._arr_np = {'logLs': logLs} if kind == 1 else {}
        self._arr_cache = (None, None)
        self.samples = _SampleList(self)
        self._core = core
        self._parameters = None if core is not None else [(np.ones(k + 1), self.cfc.logp_uniform(k))]
        self._As = np.empty((0, k + 1))
        self._heads = np.empty((0, self.cfc.n))
        self._tables = np.empty((0, k * self.cfc.n ** 2))
        return self

    @property
    def parameters(self):
        """ the proposals used so far, [(a, logp), ...] (bild/amis.py:650-653); an adopted sampler fetches them when asked """
        if self._parameters is None:
            from . import _lib
            Q = int(_lib.lib().bild_amis_num_proposals(self._core._h))
            self._parameters = [self._core.params(q) for q in range(Q)]
        return self._parameters

    @parameters.setter
    def parameters(self, value):
        self._parameters = value

    def _new_core(self, on_device=None):
        from . import _lib
        self._core = _lib.AmisCore(self.model.transitions, self.parameters[0][0], self.parameters[0][1],
                                   self.brakes[0], self.brakes[1], self.logprior)
        if on_device is None:
            self._place_core()
        return self._core

    def _place_core(self):
        """ pooled samples to HBM (bild_amis_use_device) where that pays; `device_bookkeeping=True` insists """
        from . import _lib
        want = getattr(self, 'device_bookkeeping', None)
        if want is None:
            want = self.N >= 2000 and self.model.nStates * (self.k + 1) <= 64 and _lib.device_count() > 0
        if want:
            self._core.use_device(True)

    # -- pickling / copying: the native core is rebuilt from the pooled arrays ---------------------------
    def __getstate__(self):
        if 'dirichlet' in self.__dict__:              # (a sampler with k >= T holds nothing)
            self._pool                                # append pending chunks
            self.parameters                           # (an adopted sampler: fetch them)
        state = dict(self.__dict__)
        if state.get('_core') is not None:
            state['_arr_np'] = dict(self._arr)
            state['_arr_cache'] = (None, None)
            state['_core'] = len(self._core) > 0      # True: restore the pool as well
        return state

    def __setstate__(self, state):
        had_core = state.get('_core')
        if 'parameters' in state:       # (pickles written before `parameters` became a property)
            state = dict(state)
            state['_parameters'] = state.pop('parameters')
        self.__dict__.update(state)
        if had_core is not None and '_core' in state:
            self.samples = _SampleList(self)
            self._core = self._new_core(on_device=False)
            if had_core:
                # (`_pool_np` itself, not the property: that one would ask the new, still empty core for the samples)
                self._core.restore(self.parameters[1:], self._pool_np['ss'], self._pool_np['thetas'], self._arr_np)
            self._place_core()
        elif 'samples' in state:
            self.samples = _SampleList(self)

    @property
    def _pool(self):
        """ pooled samples ('ss', 'thetas' [, lookup data of the NumPy path]); pending chunks are appended on access """
        if ((getattr(self, '_device_drawn', 0) or getattr(self, '_adopted', False)) and self._core is not None
                and (self._pool_np is None or len(self._pool_np['ss']) + sum(len(c[0]) for c in self._chunks) != len(self._core))):
            ss, thetas = self._core.pool_samples()      # drawn on the device / by the inference driver: fetched now
            self._pool_np = {'ss': ss, 'thetas': thetas}
            self._chunks = []
            return self._pool_np
        if self._chunks:
            parts_ss = ([self._pool_np['ss']] if self._pool_np else []) + [c[0] for c in self._chunks]
            parts_th = ([self._pool_np['thetas']] if self._pool_np else []) + [c[1] for c in self._chunks]
            self._pool_np = {'ss': np.concatenate(parts_ss), 'thetas': np.concatenate(parts_th)}
            self._chunks = []
        return self._pool_np

    @_pool.setter
    def _pool(self, value):
        self._pool_np = value

    @property
    def _arr(self):
        """ pooled per-sample arrays; with the native core they are fetched when somebody looks """
        if self._core is None or len(self._core) == 0:
            return self._arr_np
        stamp = len(self._sizes)
        if self._arr_cache[0] != stamp:
            self._arr_cache = (stamp, {key: self._core.pool(key) for key in self._core.POOL})
        return self._arr_cache[1]

    @_arr.setter
    def _arr(self, value):
        self._arr_np = value

    # -- profile encoding -----------------------------------------------------------------
    def st2profile(self, s, theta):
        """ (s, theta) -> Loopingprofile (bild/amis.py:670-695) """
        T = len(self.traj)
        states = theta[0] * np.ones(T)
        if len(s) > 1:
            switches = switch_indices(np.asarray(s, dtype=np.float64)[None, :], T)[0]
            for i in range(1, len(switches)):
                states[switches[i - 1]:switches[i]] = theta[i]
            states[switches[-1]:] = theta[-1]
        return Loopingprofile(states)

    # -- densities --------------------------------------------------------------------------
    def log_proposal(self, parameters, ss, thetas):
        """
        log density of the product proposal (bild/amis.py:697-715); zero where the trace has probability zero,
        also at a pole of the Dirichlet factor (see `step`)
        """
        cont, disc = self.dirichlet.logpdf(parameters[0], ss), self.cfc.logpmf(parameters[1], thetas)
        with np.errstate(invalid='ignore'):
            return np.where(disc == -np.inf, -np.inf, cont + disc)

    def logL(self, ss, thetas):
        """
        Model likelihood of a batch (bild/amis.py:717-739): ss (N, k+1) float, thetas (N, k+1)
        int -> (N,) float64.
        """
        ss = np.asarray(ss, dtype=np.float64)
        thetas = np.asarray(thetas)
        if hasattr(self.model, 'logL_st_batch'):
            return np.asarray(self.model.logL_st_batch(ss, thetas, self.traj), dtype=np.float64)
        elif hasattr(self.model, 'logL_st'):
            return np.array([self.model.logL_st(s, theta, self.traj) for s, theta in zip(ss, thetas)])
        else:
            return np.array([self.model.logL(self.st2profile(s, theta), self.traj)
                             for s, theta in zip(ss, thetas)])

    # -- exhaustive evaluation ----------------------------------------------------------------
    def fix_exhaustive(self):
        """
        Evaluate every profile when there are few enough (bild/amis.py:741-803); the evidence
        is then exact (``dlogev`` is set to 1e-10) and the sampler is marked exhausted.
        """
        T = len(self.traj)
        Nmax = min(self.max_fcomplete, self.max_fev)
        Nprofiles = self.cfc.N_total(self.k)
        for i in range(self.k):
            Nprofiles *= T - i - 1
            if Nprofiles > Nmax:
                raise self.ExhaustionImpractical(
                    f"Parameter space too large for exhaustive sampling (number of profiles = {Nprofiles} > Nmax = {Nmax})")

        # switch positions at half-integer frames, as fractions of the trajectory
        combos = np.array(list(itertools.combinations(np.arange(T - 1) + 0.5, self.k))) / (T - 1)  # (n_ss, k)
        edges = np.concatenate([np.zeros((len(combos), 1)), combos, np.ones((len(combos), 1))], axis=1)
        ss_unique = np.diff(edges, axis=1)
        thetas_unique = self.cfc.full_sample(self.k, Nmax=Nmax)

        n_ss = len(ss_unique)
        ss = np.tile(ss_unique, (len(thetas_unique), 1))
        thetas = np.repeat(thetas_unique, n_ss, axis=0)

        sample = {'ss': ss, 'thetas': thetas}
        sample['logLs'] = self.logL(ss, thetas)
        self._pool = {'ss': ss, 'thetas': thetas}
        self._arr = {'logLs': sample['logLs']}
        self._sizes = [len(ss)]

        # evidence = mean likelihood under the (uniform) prior ensemble; KL(posterior || prior)
        top = np.max(sample['logLs'])
        with np.errstate(under='ignore'):
            rel = np.exp(sample['logLs'] - top)
        ev = np.mean(rel)
        logev = np.log(ev) + top
        with np.errstate(under='ignore'):
            KL = np.mean(sample['logLs'] * rel) / ev - logev
        self.evidences.append((logev, 1e-10, KL))
        self.exhausted = True

    # -- one AMIS iteration ---------------------------------------------------------------------
    def step(self):
        """
        One AMIS iteration (bild/amis.py:805-906).  Returns ``False`` (and does nothing) when
        the sampler is exhausted, else ``True``.
        """
        if self.exhausted:
            return False

        a_cur, logp_cur = self.parameters[-1]
        if self._core is not None:
            return self._step_native(a_cur)

        # The bookkeeping below is the reference's, evaluated on pooled arrays: one call per quantity
        # and step instead of one per earlier sample / earlier proposal (with the likelihood on the GPU
        # this host-side part is what an AMIS step costs).
        # Per-sample quantities that do not change are kept with the pool (logarithms of the intervals,
        # lookup codes of the traces), per-proposal ones with the proposals (lookup tables).
        for par in self.parameters[len(self._As):]:
            head, table = self.cfc.lookup_tables(par[1][None])
            self._As = np.concatenate([self._As, np.asarray(par[0], dtype=np.float64)[None]])
            self._heads = np.concatenate([self._heads, head])
            self._tables = np.concatenate([self._tables, table])

        def log_proposals(which, pool):
            """ (len(which), N) log densities of the proposals ``which`` at the samples of ``pool`` """
            cont = self.dirichlet.logpdf_many(self._As[which], pool['ss'], pool['log_ss'])
            disc = self.cfc.logpmf_many(None, None, pool['codes'], (self._heads[which], self._tables[which]))
            with np.errstate(invalid='ignore'):
                both = cont + disc
            # a trace of probability zero has proposal density zero, also at a pole of the Dirichlet factor (s_i = 0
            # where a_i < 1: the reference adds +inf and -inf to NaN there, which ends in "Iteration did not converge";
            # proposals that collapse onto the boundary of the simplex, a_i ~ 1e-16, produce such samples in numbers)
            return np.where(disc == -np.inf, -np.inf, both)

        pool, arr = self._pool, self._arr
        # 1. the mixture denominator of every earlier sample gains the current proposal
        if self._sizes:
            cur_old = log_proposals(slice(-1, None), pool)[0]
            with np.errstate(under='ignore'):
                logd_old = np.logaddexp(arr['logδs'], cur_old)

        # 2. the new sample and its own denominator: all proposals used so far
        new_ss = self.dirichlet.sample(a_cur, self.N)
        new_thetas = self.cfc.sample(logp_cur, self.N)
        new_logLs = self.logL(new_ss, new_thetas)
        fresh = {'ss': new_ss, 'thetas': new_thetas, 'ssT': np.ascontiguousarray(new_ss.T),
                 'log_ss': self.dirichlet.log_samples(new_ss), 'codes': self.cfc.trace_codes(new_thetas)}
        per_proposal = log_proposals(slice(None), fresh)                                  # (steps, N)
        new_logd = logsumexp(per_proposal, axis=0)
        if not self._sizes:
            self._pool = fresh
            arr = {'logLs': new_logLs, 'logδs': new_logd, 'cur_log_proposal': per_proposal[-1]}
        else:
            n_old = len(pool['ss'])
            self._pool = {
                'ss': np.concatenate([pool['ss'], fresh['ss']]),
                'thetas': np.concatenate([pool['thetas'], fresh['thetas']]),
                'ssT': np.concatenate([pool['ssT'], fresh['ssT']], axis=1),
                'log_ss': (np.concatenate([pool['log_ss'][0], fresh['log_ss'][0]], axis=1),
                           np.concatenate([pool['log_ss'][1], fresh['log_ss'][1] + n_old])),
                'codes': (np.concatenate([pool['codes'][0], fresh['codes'][0]]),
                          np.concatenate([pool['codes'][1], fresh['codes'][1]])),
            }
            arr = {'logLs': np.concatenate([arr['logLs'], new_logLs]),
                   'logδs': np.concatenate([logd_old, new_logd]),
                   'cur_log_proposal': np.concatenate([cur_old, per_proposal[-1]])}
        self._sizes.append(len(new_ss))

        # 3. deterministic-mixture weights: L / mean over proposals
        arr['log_weights'] = arr['logLs'] - arr['logδs'] + np.log(len(self.parameters))
        self._arr = arr
        pooled = dict(arr, ss=self._pool['ss'], thetas=self._pool['thetas'])

        # refit, then brake
        new_a = self.dirichlet.estimate(pooled['ss'], pooled['log_weights'], self._pool['ssT'])
        new_logp = self.cfc.estimate(pooled['thetas'], pooled['log_weights'])

        limit_c = self.N * self.brakes[0]
        log_ratio = np.log(np.sum(new_a) / np.sum(a_cur))
        if np.abs(log_ratio) > limit_c:
            new_a = new_a * np.exp(np.sign(log_ratio) * limit_c - log_ratio)

        limit_p = self.N * self.brakes[1]
        with np.errstate(under='ignore'):
            p_old = np.exp(logp_cur)
            p_new = np.exp(new_logp)
        for i in range(p_new.shape[1]):
            delta = p_new[:, i] - p_old[:, i]
            biggest = np.max(np.abs(delta))
            if biggest > limit_p:
                with np.errstate(divide='ignore'):
                    new_logp[:, i] = np.log(p_old[:, i] + limit_p * delta / biggest)

        self.parameters.append((new_a, new_logp))

        # evidence, its standard error, and KL(posterior || current proposal)
        top = np.max(pooled['log_weights'])
        with np.errstate(under='ignore'):
            rel = np.exp(pooled['log_weights'] - top)
        rel[rel < np.finfo(np.float64).tiny] = 0.   # subnormal weights: no effect on the sums, slow arithmetic
        ev = np.mean(rel)
        logev = np.log(ev) + top + self.logprior
        dlogev = np.std(rel, ddof=1) / np.sqrt(len(rel)) / ev      # standard error of the mean
        with np.errstate(under='ignore', invalid='ignore'):
            # zero-weight samples with cur_log_proposal = -inf give nan terms: dropped from the
            # sum but kept in the normalisation (bild/amis.py:883-898)
            KL = (np.nansum(rel * (pooled['logLs'] - pooled['cur_log_proposal'])) / len(rel) / ev
                  - logev + self.logprior)
        self.evidences.append((logev, dlogev, KL))

        if (len(self._sizes) + 1) * self.N >= self.max_fev:
            self.exhausted = True
        return True

    def _step_native(self, a_cur):
        """
        `step` with the bookkeeping in native code.  The random numbers are drawn here, from the global NumPy
        stream in the reference's order: the Dirichlet draws, then what ``np.random.choice`` consumes for the
        first state of every trace (one uniform number each), then ``np.random.rand(N, 1)`` per later slot --
        as one block, which is the same stream.
        """
        if self.rng == 'device' and self._fusable() and not self._chunks and (self._pool_np is None or self._device_drawn):
            # draws, likelihood and bookkeeping on the device: nothing goes up but the proposal, nothing comes down but
            # partial sums; the samples are fetched when somebody looks (`_pool`)
            evidence = self._core.step_device_rng(self.model.handle(), self.model.trajset(self.traj), self.N, self.seed,
                                                  path=self.model.path)
            self._device_drawn += self.N
            self._sizes.append(self.N)
            self.parameters.append(self._core.params(-1))
            self.evidences.append(evidence)
            if (len(self._sizes) + 1) * self.N >= self.max_fev:
                self.exhausted = True
            return True
        new_ss = self.dirichlet.sample(a_cur, self.N)
        new_thetas = self._core.sample_traces(np.random.random_sample((self.k + 1, self.N)))
        self._chunks.append((new_ss, new_thetas))   # pooled on demand (`_pool`): a step itself does not need them
        self._sizes.append(len(new_ss))
        if self._fusable():
            # likelihood and bookkeeping in ONE native call: the samples go up once, the log-likelihoods stay in HBM with
            # the pooled samples (`_arr` fetches them when somebody looks), a few hundred partial sums come down
            evidence = self._core.step_fused(self.model.handle(), self.model.trajset(self.traj), new_ss, new_thetas,
                                             path=self.model.path)
        else:
            new_logLs = self.logL(new_ss, new_thetas)
            evidence = self._core.step(new_ss, new_thetas, new_logLs)    # RuntimeError if the CFC fit does not converge
        self.parameters.append(self._core.params(-1))
        self.evidences.append(evidence)
        if (len(self._sizes) + 1) * self.N >= self.max_fev:
            self.exhausted = True
        return True

    def _fusable(self):
        """
        the fused step applies when the pooled samples are on the device, the likelihood is this package's GPU kernel on
        this very model (not a wrapper around it, not an overridden `logL`) and the lists fit the walk kernel (k + 1 <= 16)
        """
        from .models import MultiStateRouse
        return (getattr(self._core, 'on_device', False) and self.fused and type(self.model) is MultiStateRouse
                and 'logL' not in self.__dict__ and type(self).logL is FixedkSampler.logL and self.k + 1 <= 16)

    # -- summaries ----------------------------------------------------------------------------------
    def tstat(self, other):
        """ evidence separation from another sampler in units of the combined standard error (bild/amis.py:908-926) """
        logev0, dlogev0 = self.evidences[-1][:2]
        logev1, dlogev1 = other.evidences[-1][:2]
        return (logev0 - logev1) / np.sqrt(dlogev0 ** 2 + dlogev1 ** 2)

    def MAP_profile(self):
        """ the sampled profile of highest likelihood (bild/amis.py:928-943) """
        j = int(np.argmax(self._arr['logLs']))   # first occurrence of the maximum, as the reference's nested argmax
        return self.st2profile(self._pool['ss'][j], self._pool['thetas'][j])

    def log_marginal_posterior(self, native=True):
        """
        (n, T) normalised log posterior marginals of the state at each frame (bild/amis.py:945-972).

        Per frame and state the weights of the samples whose profile is in that state are summed -- sums of
        non-negative terms only, like the reference's logsumexp, so that tiny marginals keep their relative
        accuracy.  ``native=True``: every interval of every sample goes into a range-add tree (O(P k log T),
        csrc/amis_host.cpp); ``native=False``: the samples are expanded to frames in vectorised NumPy chunks
        (O(P T): seconds for a finished sampler; kept as the specification).
        """
        pooled = dict(self._arr, ss=self._pool['ss'], thetas=self._pool['thetas'])
        log_weights = pooled['log_weights'] if 'log_weights' in pooled else pooled['logLs']
        T = len(self.traj)
        n = self.model.nStates
        thetas = np.asarray(pooled['thetas'])
        k1 = thetas.shape[1]
        top = np.max(log_weights)
        with np.errstate(under='ignore'):
            w = np.exp(log_weights - top)
        if native:
            from . import _lib
            seg_start, seg_state = segments_from_st(pooled['ss'], thetas, T)
            post = _lib.interval_marginals(seg_start, seg_state, w, n, T)
        else:
            starts = switch_indices(pooled['ss'], T) if k1 > 1 else np.zeros((len(thetas), 0), dtype=np.int32)
            frames = np.arange(T)[None, :]
            post = np.zeros((n, T))
            chunk = max(1, (1 << 22) // max(T, 1))
            for lo in range(0, len(thetas), chunk):
                hi = min(lo + chunk, len(thetas))
                states = np.repeat(thetas[lo:hi, :1], T, axis=1)
                for i in range(1, k1):                                   # later intervals overwrite, as st2profile does
                    states = np.where(frames >= starts[lo:hi, i - 1:i], thetas[lo:hi, i:i + 1], states)
                for st in range(n):
                    post[st] += w[lo:hi] @ (states == st)
        with np.errstate(divide='ignore', under='ignore'):
            logpost = np.log(post) + top
            return logpost - logsumexp(logpost, axis=0)
