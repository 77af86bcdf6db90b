"""
Inference models: the interface (`MultiStateModel`) and the GPU-backed multi-state Rouse
model (`MultiStateRouse`).

Counterpart of reference bild/models.py:24-370 for the one hot path this package
accelerates.  `MultiStateRouse` keeps the reference's constructor, attributes
(``models``, ``measurement``, ``localization_error``, ``transitions``, ``nStates``, ``d``)
and ``logL(profile, traj) -> float`` contract, so it can be handed to
``bild.amis.FixedkSampler`` / ``bild.core.sample`` / ``bild.postproc`` unchanged; in
addition it offers batched entry points that evaluate a whole AMIS step in one launch.

The likelihood itself runs on the GPU through the C ABI in include/bild_amd.h; there is no
CPU implementation in this package.
"""
import abc
from collections import OrderedDict

import numpy as np

from . import _lib
from . import rouse
from .profiles import Loopingprofile, segments_from_states
from .trajectory import Trajectory, as_array


class MultiStateModel(metaclass=abc.ABCMeta):
    """
    Interface used by the samplers (reference bild/models.py:24-160).

    Attributes
    ----------
    transitions : (n, n) bool -- ``transitions[i, j]``: is the switch i -> j allowed
    """

    def init_transitions(self, n):
        self.transitions = ~np.eye(n, dtype=bool)

    @property
    def nStates(self):
        return self.transitions.shape[0]

    @property
    def d(self):
        raise NotImplementedError  # pragma: no cover

    def initial_loopingprofile(self, traj):
        return Loopingprofile(np.random.choice(self.nStates, size=len(traj)))

    @abc.abstractmethod
    def logL(self, loopingprofile, traj):
        raise NotImplementedError  # pragma: no cover


class MultiStateRouse(MultiStateModel):
    """
    Multi-state Rouse model with a Kalman-filter likelihood evaluated on the GPU.

    Parameters are those of reference bild/models.py:222-228:

    N : int -- number of monomers
    D, k : float -- Rouse parameters
    d : int -- spatial dimension (1..8)
    looppositions : tuple -- per state ``None`` (no extra bond), ``(i, j[, rel_strength])`` or a
        list of such
    measurement : "end2end" or (N,) array
    localization_error : float, (d,) array or None (then ``traj.localization_error`` is used)

    Extra keyword:
    path : 'auto' | 'modal' | 'dense' -- kernel path (see include/bild_amd.h)
    """

    def __init__(self, N, D, k, d=3,
                 looppositions=(None, (0, -1)),
                 measurement="end2end",
                 localization_error=None,
                 path='auto',
                 ):
        self._d = d

        if str(measurement) == "end2end":
            measurement = np.zeros(N)
            measurement[0] = -1
            measurement[-1] = 1
        measurement = np.asarray(measurement, dtype=np.float64)
        assert len(measurement) == N
        self.measurement = measurement

        if localization_error is not None and np.isscalar(localization_error):
            localization_error = localization_error * np.ones(d)
        self.localization_error = localization_error

        self.models = []
        for loop in looppositions:
            if loop is not None and np.isscalar(loop[0]):
                loop = [loop]
            self.models.append(rouse.Model(N, D, k, d, add_bonds=loop))

        self.init_transitions(len(self.models))

        self.path = path
        self._handle = None
        self._trajsets = OrderedDict()  # small LRU of device-resident trajectory sets

    # ------------------------------------------------------------------ interface
    @property
    def d(self):
        return self._d

    def _get_noise(self, traj):
        # precedence model > trajectory > error: reference bild/models.py:255-263
        if self.localization_error is not None:
            return np.asarray(self.localization_error)
        elif getattr(traj, 'localization_error', None) is not None:
            return np.asarray(traj.localization_error)
        else:
            raise ValueError("No localization error specified (use MultiStateModel.localization_error "
                             "or Trajectory.localization_error)")

    # ------------------------------------------------------------------ device state
    @classmethod
    def from_arrays(cls, B, G, Sig, M0, C0, measurement, localization_error=None, path='auto'):
        """
        Build directly from per-state arrays (e.g. taken from an installed ``rouse``:
        ``m._dynamics['B'|'G'|'Sig']`` and ``m.steady_state()``), bypassing this package's
        own Rouse matrix builder.
        """
        self = cls.__new__(cls)
        G = np.asarray(G, dtype=np.float64)
        S, N, d = G.shape
        self._d = d
        self.measurement = np.asarray(measurement, dtype=np.float64)
        if localization_error is not None and np.isscalar(localization_error):
            localization_error = localization_error * np.ones(d)
        self.localization_error = localization_error
        self.models = None
        self._arrays = dict(B=B, G=G, Sig=Sig, M0=M0, C0=C0)
        self.init_transitions(S)
        self.path = path
        self._handle = None
        self._trajsets = OrderedDict()
        return self

    @classmethod
    def from_reference(cls, ref_model, path='auto'):
        """
        Build from a reference ``bild.models.MultiStateRouse`` (or anything with its attribute surface:
        ``models[i]._dynamics['B'|'G'|'Sig']``, ``.check_dynamics()``, ``.steady_state()``, ``measurement``,
        ``localization_error``; reference bild/src/MSRouse_logL.pyx:150-160): the matrices of the installed
        ``rouse`` package are used as they are, bypassing this package's own Rouse builder.
        """
        for mod in ref_model.models:
            mod.check_dynamics()
        steady = [mod.steady_state() for mod in ref_model.models]
        return cls.from_arrays(B=np.array([mod._dynamics['B'] for mod in ref_model.models]),
                               G=np.array([mod._dynamics['G'] for mod in ref_model.models]),
                               Sig=np.array([mod._dynamics['Sig'] for mod in ref_model.models]),
                               M0=np.array([st[0] for st in steady]), C0=np.array([st[1] for st in steady]),
                               measurement=ref_model.measurement,
                               localization_error=getattr(ref_model, 'localization_error', None), path=path)

    def __getstate__(self):
        # device handles are not state: they are recreated on first use after unpickling / copying
        state = dict(self.__dict__)
        state['_handle'] = None
        state['_trajsets'] = OrderedDict()
        return state

    def arrays(self):
        """ stacked (B, G, Sig, M0, C0) over states, as the kernel consumes them (pyx:152-163) """
        if self.models is None:
            return self._arrays
        return rouse.stack_dynamics(self.models)

    def handle(self):
        if self._handle is None:
            a = self.arrays()
            self._handle = _lib.ModelHandle(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], self.measurement)
        return self._handle

    def invalidate(self):
        """ call after changing ``models`` / ``measurement`` in place """
        self._handle = None
        self._trajsets.clear()

    def trajset(self, trajs, expect=None):
        """
        Device-resident set of trajectories (uploaded once, reused across AMIS steps).

        trajs : a trajectory or a list of trajectories
        expect : optional, the number of evaluations the set will see in total (`bild_trajset_expect`): a few single
            evaluations per trajectory are cheaper without the set's tables.  The declaration decides which tables the set
            builds (none below 300 evaluations, prefix + transient tables below 3000, all of them above or when nothing is
            declared -- the transient state table up to 4 GB, up to 64 GB from 1e8 evaluations on) and is therefore part of
            the cache key: a set declared for ten evaluations is never handed to an AMIS
            run (which asks without a declaration), and the other way round.

        The cache is keyed by the IDENTITY of the trajectory objects, and an entry is trusted while the address and shape
        of each trajectory's data, the localization errors in force and a content guard (the sum of the bit patterns of
        all values; a strided subsample of 65 536 of them for longer data) are unchanged: a lookup costs microseconds.
        In-place edits (masking frames with NaN, rescaling, refilling a preallocated array, changing single values)
        therefore lead to a fresh upload and fresh tables; only for data of more than 65 536 values can an edit between
        the guard's sample points go unnoticed -- call `invalidate()` after editing such data in place.  Trajectory-likes
        whose ``t[:]`` builds a new array on every access are keyed by a hash of their contents instead, so that they are
        not uploaded again on every call.
        """
        single = not isinstance(trajs, (list, tuple))
        items = (trajs,) if single else tuple(trajs)
        prints, arrs = self._fingerprints(items)
        if expect is not None and expect < 0:
            raise ValueError("expect must be a non-negative number of evaluations")
        # (api.cpp: kExpectPrefix, kExpectPairs, and the budget of the transient state table: 64 instead of 4 GB from 1e8 on)
        table_class = 2 if expect is None else (3 if expect >= 10 ** 8 else 2 if expect >= 3000 else 1 if expect >= 300 else 0)
        key = (table_class,) + tuple(id(t) if p[0] is not None else p for t, p in zip(items, prints))
        hit = self._trajsets.get(key)
        if hit is not None:
            ts, kept, old_prints = hit
            if all(a is b or p[0] is None for a, b, p in zip(kept, items, prints)) and old_prints == prints:
                self._trajsets.move_to_end(key)
                return ts
        noises = [np.frombuffer(p[2], dtype=np.float64) for p in prints]
        ts = _lib.TrajSetHandle(self.handle(), arrs, np.stack(noises))
        if expect is not None:
            ts.expect(expect)
        self._trajsets[key] = (ts, items, prints)   # `items` keeps the objects (and their ids) alive
        while len(self._trajsets) > 8:
            self._trajsets.popitem(last=False)
        return ts

    def _fingerprints(self, items):
        """ per trajectory: (data address or None, shape, noise bytes, content guard); and the data arrays themselves """
        out, arrs = [], []
        for t in items:
            view = t[:]
            a = view if (type(view) is np.ndarray and view.dtype == np.float64 and view.ndim == 2 and view.flags.c_contiguous) \
                else as_array(t)
            arrs.append(a)
            noise = np.ascontiguousarray(self._get_noise(t), dtype=np.float64).tobytes()
            addr = _lib.aptr(view) if isinstance(view, np.ndarray) else None
            # `t[:]` of an ndarray or of this package's Trajectory is a view of one buffer; anything else is asked twice
            stable = addr is not None and (type(t) in (np.ndarray, Trajectory) or
                                           (isinstance(t[:], np.ndarray) and t[:].__array_interface__['data'][0] == addr))
            if stable:
                # the guard: the sum of the bit patterns -- every value takes part (NaNs included, which an ordinary sum would
                # drown in); beyond 65 536 values a strided subsample of that many
                bits = a.reshape(-1).view(np.uint64)
                if bits.size > 65536:
                    bits = bits[::bits.size // 65536 + 1]
                out.append((addr, a.shape, noise, int(np.add.reduce(bits))))
            else:   # no stable buffer to identify the data by: the contents are the key
                out.append((None, a.shape, noise, hash(a.tobytes())))
        return out, arrs

    # ------------------------------------------------------------------ likelihood
    def logL(self, profile, traj):
        """
        log p(traj | profile, model), reference bild/models.py:265-278 -> MSRouse_logL.

        Returns
        -------
        float
        """
        states = np.asarray(profile[:])
        assert len(states) == len(traj)
        return float(_lib.logl_profiles(self.handle(), self.trajset(traj), states[None, :], path=self.path)[0])

    def logL_batch(self, profiles, traj):
        """ many expanded profiles on one trajectory: (n, T) int array or list of Loopingprofile -> (n,) """
        if not isinstance(profiles, np.ndarray):
            profiles = np.stack([np.asarray(p[:]) for p in profiles])
        seg_start, seg_state = segments_from_states(profiles)
        return _lib.logl_segments(self.handle(), self.trajset(traj), seg_start, seg_state, path=self.path)

    def logL_st_batch(self, ss, thetas, traj):
        """
        One AMIS batch in (s, theta) parametrisation: what reference
        ``FixedkSampler.logL(ss, thetas)`` (bild/amis.py:717-739) computes with a Python loop.
        """
        return _lib.logl_st(self.handle(), self.trajset(traj), ss, thetas, path=self.path)

    def logL_st_batch_to_device(self, ss, thetas, traj, d_out, stream=0):
        """
        Same batch, results left in HBM at the raw device pointer ``d_out`` (float64[len(thetas)]), asynchronous on
        the HIP stream ``stream``: what `dist.ShardedModel` feeds into the all-gather of a multi-GPU AMIS step.
        """
        _lib.logl_st_to_device(self.handle(), self.trajset(traj), ss, thetas, d_out, stream=stream, path=self.path)

    def check_st_rows(self):
        """
        `logL_st_batch_to_device` waits for nothing and cannot refuse a row that is no point on the simplex (it gets NaN):
        this waits for the pending calls and raises if one of their rows was refused (bild_logl_st_status)
        """
        _lib.logl_st_status(self.handle())

    def logL_st(self, s, theta, traj):
        """ the per-sample hook the reference sampler prefers when present (bild/amis.py:734-736) """
        return float(self.logL_st_batch(np.asarray(s)[None, :], np.asarray(theta)[None, :], traj)[0])

    def logL_segments(self, seg_start, seg_state, trajs, traj_id=None):
        """ general batch: run-length encoded profiles over a set of trajectories """
        return _lib.logl_segments(self.handle(), self.trajset(trajs), seg_start, seg_state, traj_id, path=self.path)

    # ------------------------------------------------------------------ generative model
    def initial_loopingprofile(self, traj):
        """ initial guess: the per-frame best state of the factorized model (reference bild/models.py:280-293) """
        return self.toFactorized().initial_loopingprofile(traj)

    def toFactorized(self):
        """
        The `FactorizedModel` that draws every frame from the steady-state distance distribution of its
        state (reference bild/models.py:352-370): Maxwell with scale^2 = w.C0.w + mean squared localization
        error per dimension.  CPU only, cheap; not a substitute for `logL`.
        """
        from scipy import stats
        err = self.localization_error
        noise2_per_d = np.sum(np.asarray(err) ** 2) / self.d if err is not None else 0
        w = self.measurement
        return FactorizedModel([stats.maxwell(scale=np.sqrt(w @ C0 @ w + noise2_per_d)) for C0 in self.arrays()['C0']],
                               d=self.d)

    def trajectory_from_loopingprofile(self, profile, localization_error=None, missing_frames=None, rng=None):
        """
        Sample a trajectory from the model (reference bild/models.py:295-350): steady-state
        conformation of ``profile[0]``, propagate with ``profile[t]``, measure, blank the
        missing frames, then add localization noise.
        """
        rng = np.random.default_rng() if rng is None else rng
        if localization_error is None:
            if self.localization_error is None:
                raise ValueError("Need to specify either localization_error or model.localization_error")
            localization_error = self.localization_error
        if np.isscalar(localization_error):
            localization_error = self.d * [localization_error]
        localization_error = np.asarray(localization_error, dtype=np.float64)
        if localization_error.shape != (self.d,):
            raise ValueError("Did not understand localization_error")

        T = len(profile)
        if missing_frames is None or (np.isscalar(missing_frames) and missing_frames == 0):
            missing = np.array([], dtype=int)
        elif np.isscalar(missing_frames):
            if 0 < missing_frames < 1:
                missing = np.nonzero(rng.random(T) < missing_frames)[0]
            else:
                missing = rng.choice(T, size=int(missing_frames), replace=False).astype(int)
        else:
            missing = np.asarray(missing_frames, dtype=int)

        data = np.full((T, self.d), np.nan)
        conf = self.models[profile[0]].conf_ss(rng)
        data[0] = self.measurement @ conf
        for i in range(1, T):
            conf = self.models[profile[i]].evolve(conf, rng)
            data[i] = self.measurement @ conf
        data[missing, :] = np.nan
        data += localization_error[None, :] * rng.standard_normal(data.shape)
        return Trajectory(data, localization_error=localization_error, loopingprofile=profile)


class FactorizedModel(MultiStateModel):
    """
    Time-scale-separated stand-in likelihood: every frame is drawn independently from the
    distance distribution of its state (reference bild/models.py:372-534).  Cheap, CPU only;
    the reference's own tests use it as the likelihood double for everything above the kernel,
    and so do this package's sampler tests.

    distributions : objects with ``logpdf(r)`` (e.g. ``scipy.stats.maxwell(scale=...)``)
    """

    def __init__(self, distributions, d=3):
        self.distributions = distributions
        self._d = d
        self._tables = {}
        self.init_transitions(len(distributions))

    @property
    def d(self):
        return self._d

    def clear_memo(self):
        self._tables = {}

    def _table(self, traj):
        key = id(traj)
        hit = self._tables.get(key)
        if hit is None or hit[0] is not traj:
            arr = as_array(traj)
            r = np.sqrt(np.sum(arr ** 2, axis=1))
            with np.errstate(divide='ignore', invalid='ignore'):
                table = np.array([dist.logpdf(r) for dist in self.distributions])  # (n, T), NaN on missing frames
            hit = (traj, table)
            self._tables[key] = hit
        return hit[1]

    def initial_loopingprofile(self, traj):
        """ per-frame maximum-likelihood state, missing frames filled from the next valid one """
        table = self._table(traj)
        valid = np.nonzero(~np.any(np.isnan(as_array(traj)), axis=1))[0]
        best = np.argmax(table[:, valid], axis=0)
        states = np.zeros(len(traj), dtype=int)
        states[:valid[0] + 1] = best[0]
        last = valid[0]
        for t, s in zip(valid[1:], best[1:]):
            states[last + 1:t + 1] = s
            last = t
        states[last + 1:] = best[-1]
        return Loopingprofile(states)

    def logL(self, profile, traj):
        table = self._table(traj)
        states = np.asarray(profile[:], dtype=int)
        return float(np.nansum(table[states, np.arange(len(states))]))

    def trajectory_from_loopingprofile(self, profile, localization_error=0., missing_frames=None):
        """
        Generative model (reference bild/models.py:487-534): a distance from the state's distribution and a
        uniformly random direction per frame.  The localization error is part of the distributions already: it is
        only recorded on the trajectory, not added.
        """
        if np.isscalar(localization_error):
            localization_error = self.d * [localization_error]
        localization_error = np.asarray(localization_error, dtype=np.float64)
        if localization_error.shape != (self.d,):
            raise ValueError("Did not understand localization_error")
        T = len(profile)
        if missing_frames is None or (np.isscalar(missing_frames) and missing_frames == 0):
            missing = np.array([], dtype=int)
        elif np.isscalar(missing_frames):
            if 0 < missing_frames < 1:
                missing = np.nonzero(np.random.rand(T) < missing_frames)[0]
            else:
                missing = np.random.choice(T, size=missing_frames, replace=False).astype(int)
        else:
            missing = np.asarray(missing_frames, dtype=int)
        magnitudes = np.array([self.distributions[state].rvs() for state in profile[:]])
        data = np.random.normal(size=(len(magnitudes), self.d))
        data *= np.expand_dims(magnitudes / np.linalg.norm(data, axis=1), 1)
        data[missing, :] = np.nan
        return Trajectory(data, localization_error=localization_error, loopingprofile=profile)

    def logL_batch(self, profiles, traj):
        table = self._table(traj)
        if not isinstance(profiles, np.ndarray):
            profiles = np.stack([np.asarray(p[:]) for p in profiles])
        return np.nansum(table[profiles, np.arange(profiles.shape[1])[None, :]], axis=1)
