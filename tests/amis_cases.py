"""
Deterministic AMIS test problems shared by the golden generator (reference side) and the
parity tests (this package's side): a table likelihood  logL(profile) = sum_t table[state_t, t]
and the sampler settings of each case.
"""
import numpy as np


class TableModel:
    """ minimal MultiStateModel duck type with a per-frame log-likelihood table (n, T) """

    def __init__(self, table, transitions=None):
        self.table = np.asarray(table, dtype=float)
        n = self.table.shape[0]
        self.transitions = ~np.eye(n, dtype=bool) if transitions is None else np.asarray(transitions, dtype=bool)
        self.d = 1

    @property
    def nStates(self):
        return self.transitions.shape[0]

    def logL(self, profile, traj):
        states = np.asarray(profile[:], dtype=int)
        return float(np.sum(self.table[states, np.arange(len(states))]))


def _table(seed, n, T, truth_switches):
    """ frames prefer the state of a piecewise-constant ground truth, with noise """
    rng = np.random.default_rng(seed)
    truth = np.zeros(T, dtype=int)
    s = 0
    for t in truth_switches:
        s = (s + 1) % n
        truth[t:] = s
    table = -1.5 * np.abs(rng.standard_normal((n, T))) - 2.0
    table[truth, np.arange(T)] += 2.5
    return table


CASES = {
    # the three regimes of reference tests/test_amis.py:220-235: exhaustive, sampled until max_fev, k >= T
    'exhaustive_k1': dict(table=_table(1, 2, 6, [2]), transitions=None, k=1, N=100, seed=11, steps=1,
                          max_fev=20000, max_fcomplete=1000),
    'sampled_k2_T30': dict(table=_table(2, 2, 30, [8, 21]), transitions=None, k=2, N=40, seed=12, steps=6,
                           max_fev=20000, max_fcomplete=10),
    'sampled_k3_3state': dict(table=_table(3, 3, 40, [10, 19, 33]), transitions=None, k=3, N=60, seed=13, steps=5,
                              max_fev=20000, max_fcomplete=10),
    'restricted_transitions': dict(table=_table(4, 3, 25, [7, 15]), transitions=[[0, 1, 1], [1, 0, 0], [1, 1, 0]],
                                   k=2, N=50, seed=14, steps=4, max_fev=20000, max_fcomplete=10),
    'exhausts_by_max_fev': dict(table=_table(5, 2, 20, [9]), transitions=None, k=2, N=10, seed=15, steps=3,
                                max_fev=25, max_fcomplete=10),
}
