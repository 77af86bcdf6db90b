"""
Greedy local optimisation of the boundaries of an inferred profile.

Counterpart of reference bild/postproc.py:13-117.  Each iteration
needs the likelihood of the current profile and of the 2k profiles obtained by moving one of
its k boundaries one frame left or right; they are evaluated as ONE batch when the model
offers ``logL_batch`` (the GPU model and `FactorizedModel` do).
"""
import numpy as np


def _batch_logL(model, profiles, traj):
    states = np.stack([np.asarray(p[:]) for p in profiles])
    if hasattr(model, 'logL_batch'):
        return np.asarray(model.logL_batch(states, traj), dtype=float)
    return np.array([model.logL(p, traj) for p in profiles])


def logLR_boundaries(profile, traj, model):
    """
    (k, 2) log likelihood ratios of moving boundary i to the left ([i, 0]) / right ([i, 1])
    (bild/postproc.py:13-59); empty array for a profile without boundaries.
    """
    boundaries = np.nonzero(np.diff(profile.state))[0]   # boundary sits between b and b+1
    if len(boundaries) == 0:
        return np.array([])
    candidates = [profile]
    for b in boundaries:
        left = profile.copy()
        left[b] = profile[b + 1]
        right = profile.copy()
        right[b + 1] = profile[b]
        candidates += [left, right]
    logLs = _batch_logL(model, candidates, traj)
    return logLs[1:].reshape(len(boundaries), 2) - logLs[0]


class BoundaryEliminationError(Exception):
    pass


def optimize_boundary(profile, traj, model, max_iteration=10000):
    """
    Repeatedly make the single one-frame boundary move that raises the likelihood most, until no
    move helps (bild/postproc.py:64-117).

    Raises `BoundaryEliminationError` if the best move would shrink an interval to nothing, and
    ``RuntimeError`` after ``max_iteration`` moves.
    """
    cur = profile.copy()
    for _ in range(max_iteration):
        logLR = logLR_boundaries(cur, traj, model)
        if len(logLR) == 0:
            break
        i, j = np.unravel_index(np.argmax(logLR), logLR.shape)
        if not logLR[i, j] > 0:
            break
        b = np.nonzero(np.diff(cur.state))[0][i]
        vanishes = ((j == 0 and (b == 0 or cur[b - 1] == cur[b + 1]))
                    or (j == 1 and (b == len(traj) - 2 or cur[b + 2] == cur[b])))
        if vanishes:
            raise BoundaryEliminationError(f"Trying to abolish boundary at {b}")
        cur[b + j] = cur[b + (1 - j)]
    else:
        raise RuntimeError(f"Exceeded max_iteration = {max_iteration}")
    return cur
