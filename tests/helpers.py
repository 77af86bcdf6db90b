"""
Shared builders for the test-suite and bench: synthetic Rouse models, trajectories and
candidate looping profiles, plus thin adapters to hand this
package's objects to the reference kernels / the CPU oracle.
"""
import numpy as np

from bild_amd import rouse
from bild_amd.profiles import switch_indices, states_from_segments, segments_from_st
from bild_amd.trajectory import Trajectory

LOOPS = {
    2: (None, (0, -1)),
    3: (None, (0, -1), (0, 10)),
}


def end2end(N):
    w = np.zeros(N)
    w[0], w[-1] = -1., 1.
    return w


class DuckModel:
    """ the attribute surface the reference kernels read (pyx:144-160, _py.py:68-84) """

    def __init__(self, N=20, D=1., k=5., d=3, loops=LOOPS[2], localization_error=None, measurement=None):
        self.d = d
        self.models = []
        for loop in loops:
            if loop is not None and np.isscalar(loop[0]):
                loop = [loop]
            self.models.append(rouse.Model(N, D, k, d, add_bonds=loop))
        self.measurement = end2end(N) if measurement is None else np.asarray(measurement, dtype=float)
        if localization_error is not None and np.isscalar(localization_error):
            localization_error = localization_error * np.ones(d)
        self.localization_error = localization_error

    def _get_noise(self, traj):
        if self.localization_error is not None:
            return np.asarray(self.localization_error)
        if getattr(traj, 'localization_error', None) is not None:
            return np.asarray(traj.localization_error)
        raise ValueError("No localization error specified")

    def arrays(self):
        return rouse.stack_dynamics(self.models)


def random_profile(rng, T, S, mean_dwell):
    """ piecewise-constant ground truth; dwell ~ Geometric(mean_dwell), never repeats a state """
    states = np.empty(T, dtype=np.int64)
    t, s = 0, int(rng.integers(S))
    while t < T:
        dwell = int(rng.geometric(1. / mean_dwell))
        states[t:t + dwell] = s
        t += dwell
        if S > 1:
            s = int((s + 1 + rng.integers(S - 1)) % S)
    return states


def synth_trajectory(model, states, err, rng, missing=None):
    """
    Generative model of reference bild/models.py:295-350: steady state of states[0],
    propagate with states[t], measure, blank missing frames, add localization noise.
    """
    T, d = len(states), model.d
    w = model.measurement
    data = np.empty((T, d))
    conf = model.models[states[0]].conf_ss(rng)
    data[0] = w @ conf
    for t in range(1, T):
        conf = model.models[states[t]].evolve(conf, rng)
        data[t] = w @ conf
    if missing is not None and len(missing):
        data[np.asarray(missing, dtype=int), :] = np.nan
    err = np.asarray(err, dtype=float) * np.ones(d)
    data += err[None, :] * rng.standard_normal(data.shape)
    return Trajectory(data, localization_error=err)


def missing_mask(rng, T, kind):
    """ 'none', 'iid' (10 %), 'bursty' (gaps ~Geom(20) covering ~30 %) """
    if kind == 'none':
        return np.array([], dtype=int)
    if kind == 'iid':
        return np.nonzero(rng.random(T) < 0.1)[0]
    if kind == 'bursty':
        mask = np.zeros(T, dtype=bool)
        while mask.mean() < 0.3:
            start = int(rng.integers(T))
            mask[start:start + int(rng.geometric(1. / 20))] = True
        return np.nonzero(mask)[0]
    raise ValueError(kind)


def candidate_profiles(rng, n, k, S, transitions=None):
    """
    AMIS-like candidates: ss ~ Dirichlet(1), thetas a random walk on the allowed transitions
    (uniform over successors: never stays in the same state).
    """
    ss = rng.dirichlet(np.ones(k + 1), size=n)
    thetas = np.empty((n, k + 1), dtype=np.int64)
    thetas[:, 0] = rng.integers(S, size=n)
    for i in range(1, k + 1):
        step = 1 + rng.integers(max(S - 1, 1), size=n)
        thetas[:, i] = (thetas[:, i - 1] + step) % S if S > 1 else 0
    return ss, thetas


def expand(ss, thetas, T):
    """ (ss, thetas) -> (n, T) expanded states with the reference's st2profile rule """
    seg_start, seg_state = segments_from_st(ss, thetas, T)
    return states_from_segments(seg_start, seg_state, T)


class ProfileView:
    """ what the reference kernels need of a Loopingprofile: [0], [:], [1:] """

    def __init__(self, states):
        self.state = np.asarray(states, dtype=int)

    def __getitem__(self, key):
        return self.state[key]

    def __len__(self):
        return len(self.state)
