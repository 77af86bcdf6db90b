#!/usr/bin/env python3
"""
Host-side cost of one AMIS step, proposal bookkeeping only (likelihood time subtracted): the reference's
bild/amis.py (imported through oracle/ref_loader.py -- build container only) against bild_amd.amis, same seed,
same table likelihood, T = 1000, k = 4.  Also checks that both walk through the same evidences.

    python tests/tools/amis_vs_reference.py [N] [steps]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
import numpy as np
import bild_amd
from bild_amd.profiles import segments_from_st
from oracle import ref_loader

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
T, k = 1000, 4
rng = np.random.default_rng(0)
true = np.zeros(T, int); true[200:450] = 1; true[700:820] = 1
table = np.where(true[None, :] == np.arange(2)[:, None], -0.5, -1.5) + 0.1 * rng.standard_normal((2, T))
cum = np.concatenate([np.zeros((2, 1)), np.cumsum(table, axis=1)], axis=1)
spent = [0.0]


class TableModel:
    transitions = ~np.eye(2, dtype=bool); nStates = 2; d = 3

    def logL_st_batch(self, ss, thetas, traj):
        t0 = time.perf_counter()
        a, b = segments_from_st(ss, thetas, T)
        ends = np.concatenate([a[:, 1:], np.full((len(a), 1), T)], axis=1)
        out = np.sum(cum[b, np.minimum(ends, T)] - cum[b, np.minimum(a, T)], axis=1)
        spent[0] += time.perf_counter() - t0
        return out


class RefSampler(ref_loader.load('amis').FixedkSampler):
    def logL(self, ss, thetas):     # the reference loops over samples in Python: give it the batch instead
        return self.model.logL_st_batch(ss, thetas, self.traj)


def run(cls):
    np.random.seed(1)
    sampler = cls(np.zeros((T, 3)), TableModel(), k=k, N=N, max_fev=10 ** 9)
    spent[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        sampler.step()
    return (time.perf_counter() - t0 - spent[0]) / steps * 1e3, np.array(sampler.evidences)


t_ref, ev_ref = run(RefSampler)
t_own, ev_own = run(bild_amd.FixedkSampler)
print(f"N={N}, {steps} steps: reference bookkeeping {t_ref:.2f} ms/step, bild_amd {t_own:.2f} ms/step "
      f"({t_ref / t_own:.1f}x); max |d logev| {np.max(np.abs(ev_ref[:, 0] - ev_own[:, 0])):.2e}")
