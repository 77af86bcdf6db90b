"""
Minimal trajectory container.

The reference uses ``noctiluca.Trajectory`` (un-vendored, absent here) purely as a
container on the likelihood path: ``traj[:]`` (T, d) float64 with NaN = missing frame,
``traj[t]``, ``len(traj)``, ``traj.localization_error``
(reference bild/src/MSRouse_logL.pyx:171-178, bild/models.py:255-263).  This class
provides exactly that surface; any object offering it (including a real
``noctiluca.Trajectory``) is accepted wherever a `Trajectory` is.
"""
import numpy as np


class Trajectory:
    def __init__(self, data, localization_error=None, loopingprofile=None):
        data = np.array(data, dtype=np.float64)
        if data.ndim == 1:
            data = data[:, None]
        assert data.ndim == 2
        self.data = np.ascontiguousarray(data)
        if localization_error is not None:
            localization_error = np.asarray(localization_error, dtype=np.float64)
            if localization_error.ndim == 0:
                localization_error = localization_error * np.ones(self.d)
        self.localization_error = localization_error
        self.meta = {'loopingprofile': loopingprofile}

    @property
    def T(self):
        return self.data.shape[0]

    @property
    def d(self):
        return self.data.shape[1]

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, key):
        return self.data[key]

    def abs(self):
        """ trajectory of the Euclidean norm per frame, (T, 1) (noctiluca ``Trajectory.abs()``) """
        return Trajectory(np.sqrt(np.sum(self.data ** 2, axis=1, keepdims=True)),
                          localization_error=None, loopingprofile=self.meta.get('loopingprofile'))

    def valid_frames(self):
        return ~np.any(np.isnan(self.data), axis=1)

    def count_valid_frames(self):
        return int(np.count_nonzero(self.valid_frames()))


def as_array(traj):
    """ (T, d) C-contiguous float64 view/copy of anything trajectory-like """
    arr = np.asarray(traj[:], dtype=np.float64)
    if arr.ndim == 1:
        arr = arr[:, None]
    return np.ascontiguousarray(arr)


def make_trajectory(traj):
    """
    Accept what reference ``bild.sample`` accepts on its first argument as far as the likelihood
    path needs it (the reference delegates to ``noctiluca.make_Trajectory``, bild/core.py:111):
    anything trajectory-like is passed through, arrays of shape (T,) or (T, d) are wrapped.
    """
    if isinstance(traj, np.ndarray) or isinstance(traj, (list, tuple)):
        arr = np.asarray(traj, dtype=np.float64)
        if arr.ndim > 2:
            raise ValueError("pass the (T, d) distance trajectory of one pair of loci")
        return Trajectory(arr)
    return traj
