"""
The AMIS batch boundary: `FixedkSampler.logL(ss, thetas)`.

Counterpart of reference bild/amis.py:623-739 restricted to what the hot path needs: the
constructor's data members, ``st2profile`` and the batch likelihood.  Where the reference
loops over the N samples of a step in Python and calls the model once per sample
(bild/amis.py:735-739), this class hands the whole batch to the model in one call when the
model offers ``logL_st_batch`` (the GPU-backed `MultiStateRouse` does), and otherwise
behaves exactly like the reference.
"""
import numpy as np

from .profiles import Loopingprofile, switch_indices


class FixedkSampler:
    """
    Holds one (trajectory, model, k) problem; evaluates batches of candidate profiles.

    Parameters follow reference bild/amis.py:623-629.
    """

    class ExhaustionImpractical(ValueError):
        pass

    def __init__(self, traj, model, k,
                 N=100,
                 concentration_brake=1e-2,
                 polarization_brake=1e-3,
                 max_fev=20000,
                 max_fcomplete=1000,
                 ):
        self.k = k
        self.N = N
        self.brakes = (concentration_brake, polarization_brake)
        self.max_fev = max_fev
        self.max_fcomplete = max_fcomplete
        self.exhausted = False
        self.traj = traj
        self.model = model
        self.samples = []
        self.evidences = []

    def st2profile(self, s, theta):
        """
        (s, theta) -> Loopingprofile, reference bild/amis.py:670-695.

        s : (k+1,) float, on the unit simplex;  theta : (k+1,) int
        """
        T = len(self.traj)
        states = theta[0] * np.ones(T)
        if len(s) > 1:
            switches = switch_indices(np.asarray(s)[None, :], T)[0]
            for i in range(1, len(switches)):
                states[switches[i - 1]:switches[i]] = theta[i]
            states[switches[-1]:] = theta[-1]
        return Loopingprofile(states)

    def logL(self, ss, thetas):
        """
        Evaluate the model likelihood for a batch (reference bild/amis.py:717-739).

        ss : (N, k+1) float64 ; thetas : (N, k+1) int  ->  (N,) float64
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
