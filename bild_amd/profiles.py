"""
Looping profiles and the (s, theta) -> switch-index encoding.

`Loopingprofile` is this build's counterpart of reference bild/util.py:6-141 (same
operators and methods; the semantics of ``profile[t]`` are documented at
bild/util.py:15-23).  `switch_indices` / `segments_from_*` produce the compact
run-length encoding the HIP kernels consume; they restate the integer arithmetic of
``FixedkSampler.st2profile`` (reference bild/amis.py:670-695) on the host, in NumPy, so
that no device floating point can disagree with the reference about where a switch falls.
"""
import numpy as np


class Loopingprofile:
    """
    Thin wrapper around an integer state array.

    ``profile[0]`` selects the steady state the trajectory starts from; ``profile[t]``
    (t >= 1) selects the model state used to propagate *to* frame ``t``.
    """

    def __init__(self, states=None):
        if states is None:
            self.state = np.array([], dtype=int)
        else:
            self.state = np.asarray(states, dtype=int)

    def copy(self):
        new = Loopingprofile()
        new.state = self.state.copy()
        return new

    def __len__(self):
        return len(self.state)

    def __getitem__(self, key):
        return self.state[key]

    def __setitem__(self, key, val):
        val = np.asarray(val)
        assert np.issubdtype(val.dtype, np.integer)
        self.state[key] = val

    def __eq__(self, other):
        try:
            if len(self) != len(other):
                return False
            return bool(np.all(self.state == other.state))
        except Exception:
            return False

    def _run_starts(self):
        """ frames at which a new run of equal states begins (frame 0 excluded) """
        return np.flatnonzero(self.state[1:] != self.state[:-1]) + 1

    def count_switches(self):
        return int(self._run_starts().size)

    def intervals(self):
        """
        Runs of equal state as (start, end, state) triples (reference bild/util.py:109-126); the open ends of the profile
        are reported as None, so that ``profile.state[start:end]`` is the run.
        """
        cuts = self._run_starts().tolist()
        lefts = [None] + cuts
        rights = cuts + [None]
        return [(lo, hi, self.state[-1 if hi is None else hi - 1]) for lo, hi in zip(lefts, rights)]

    def plottable(self):
        """
        Staircase (t, y) for plotting against the frame axis (reference bild/util.py:128-141): every run contributes its two
        end points, ``profile[t]`` being drawn between frames t-1 and t.
        """
        cuts = self._run_starts()
        edges = np.concatenate(([0], cuts, [len(self)]))
        t = np.repeat(edges, 2)[1:-1] - 1
        y = np.repeat(self.state[edges[:-1]], 2)
        return t, y


def state_probabilities(profiles, nStates=None):
    """ marginal state probabilities (nStates, T) of an ensemble of profiles """
    allstates = np.array([profile[:] for profile in profiles])
    if nStates is None:
        nStates = np.max(allstates) + 1
    counts = np.array([np.count_nonzero(allstates == i, axis=0) for i in range(nStates)])
    return counts / allstates.shape[0]


# ----------------------------------------------------------------------------------------
# compact profile encodings
# ----------------------------------------------------------------------------------------
def switch_indices(ss, T):
    """
    Switch frames of a batch of interval vectors, exactly as the reference computes them.

    reference bild/amis.py:685-688:
        ``switches = floor(cumsum(s)[:-1] * (T-1)).astype(int) + 1``

    Parameters
    ----------
    ss : (n, k+1) float64, rows on the unit simplex
    T : int, trajectory length in frames

    Returns
    -------
    (n, k) int32, each entry in [1, T]
    """
    ss = np.asarray(ss, dtype=np.float64)
    if ss.ndim == 1:
        ss = ss[None, :]
    if ss.shape[1] <= 1:
        return np.zeros((ss.shape[0], 0), dtype=np.int32)
    # np.cumsum along a row is the same sequential left-to-right sum the reference
    # performs per sample
    switchpos = np.cumsum(ss, axis=1)[:, :-1]
    return (np.floor(switchpos * (T - 1)).astype(np.int64) + 1).astype(np.int32)


def segments_from_st(ss, thetas, T):
    """
    (s, theta) batch -> run-length segments ``(seg_start, seg_state)``, both (n, k+1) int32.

    Segment ``i`` of sample ``r`` covers frames ``seg_start[r, i] <= t < seg_start[r, i+1]``
    (``T`` for the last one) and is in state ``seg_state[r, i]``.  Empty segments (equal
    switch indices, or a switch index of ``T``) are legal and are skipped by the kernels,
    which reproduces the slice-assignment semantics of reference bild/amis.py:690-693.
    """
    thetas = np.asarray(thetas)
    if thetas.ndim == 1:
        thetas = thetas[None, :]
    n, k1 = thetas.shape
    seg_start = np.zeros((n, k1), dtype=np.int32)
    if k1 > 1:
        seg_start[:, 1:] = switch_indices(ss, T)
    seg_state = np.ascontiguousarray(thetas, dtype=np.int32)
    return seg_start, seg_state


def segments_from_states(states):
    """
    Expanded profiles (n, T) -> run-length segments padded to the longest run count.

    Returns
    -------
    seg_start, seg_state : (n, K1) int32 ; padding segments start at T (empty)
    """
    states = np.asarray(states)
    if states.ndim == 1:
        states = states[None, :]
    n, T = states.shape
    change = np.ones((n, T), dtype=bool)
    change[:, 1:] = states[:, 1:] != states[:, :-1]
    nseg = change.sum(axis=1)
    K1 = int(nseg.max()) if n > 0 else 1
    seg_start = np.full((n, K1), T, dtype=np.int32)
    seg_state = np.zeros((n, K1), dtype=np.int32)
    rows, cols = np.nonzero(change)
    pos = (np.cumsum(change, axis=1) - 1)[rows, cols]
    seg_start[rows, pos] = cols
    seg_state[rows, pos] = states[rows, cols]
    return seg_start, seg_state


def states_from_segments(seg_start, seg_state, T):
    """ inverse of the encodings above: (n, K1) segments -> expanded (n, T) int64 states """
    seg_start = np.asarray(seg_start)
    seg_state = np.asarray(seg_state)
    n, K1 = seg_start.shape
    out = np.empty((n, T), dtype=np.int64)
    for r in range(n):
        out[r, :] = seg_state[r, 0]
        for i in range(1, K1):
            out[r, seg_start[r, i]:] = seg_state[r, i]
    return out
