"""
Seeded randomised GPU parity sweep: chain length, dimensions, number of states, loop positions,
measurement vector, localization errors (equal / distinct), missing-frame pattern, trajectory length,
number of switches, kernel path and reduction on/off are all drawn at random; every evaluation is
compared with the CPU oracle.  |delta logL| < 1e-8.
"""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-8


def _random_case(seed):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([3, 5, 6, 9, 10, 12, 14, 18, 20, 22, 26, 30, 36, 44, 70]))   # the last three: LDS-resident kernel
    d = int(rng.integers(1, 4))
    S = int(rng.integers(1, 4))
    loops = [None]
    while len(loops) < S:
        i, j = sorted(rng.choice(N, 2, replace=False).tolist())
        loops.append((int(i), int(j), float(rng.choice([1.0, 0.5, 2.0]))))
    if rng.random() < 0.5:
        w = H.end2end(N)
    else:
        w = np.zeros(N)
        idx = rng.choice(N, 2, replace=False)
        w[idx[0]], w[idx[1]] = -1.0, float(rng.choice([1.0, 0.7]))
    kind = rng.integers(3)
    err = [np.full(d, 0.1), 0.05 + 0.3 * rng.random(d), np.repeat(0.05 + 0.3 * rng.random(), d)][kind][:d]
    return dict(rng=rng, N=N, d=d, S=S, loops=tuple(loops), w=w, err=np.asarray(err, dtype=float),
                D=float(rng.choice([0.5, 1.0, 2.0])), k=float(rng.choice([0.5, 2.0, 5.0])),
                T=int(rng.integers(2, 260)), nsw=int(rng.integers(0, 7)), miss=str(rng.choice(['none', 'iid', 'bursty'])),
                reduce=bool(rng.integers(2)), path=str(rng.choice(['modal', 'dense'])) if N <= 32 else 'modal')


# BILD_FUZZ_SEEDS=<n> widens the sweep for a soak run
@pytest.mark.parametrize('seed', range(int(os.environ.get('BILD_FUZZ_SEEDS', '48'))))
def test_random_configuration(built_lib, seed):
    import bild_amd
    from bild_amd import _lib
    from oracle import oracle
    c = _random_case(1000 + seed)
    rng = c['rng']
    model = bild_amd.MultiStateRouse(c['N'], c['D'], c['k'], d=c['d'], looppositions=c['loops'], measurement=c['w'],
                                     localization_error=c['err'], path=c['path'])
    if not c['reduce']:
        a = model.arrays()
        model._handle = _lib.ModelHandle(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], model.measurement, reduce=False)
    T = c['T']
    truth = H.random_profile(rng, T, c['S'], max(T // 4, 1))
    miss = H.missing_mask(rng, T, c['miss']) if T > 8 else np.array([], dtype=int)
    traj = model.trajectory_from_loopingprofile(truth, missing_frames=miss, rng=rng)
    n = int(rng.integers(1, 90))
    ss, thetas = H.candidate_profiles(rng, n, c['nsw'], c['S'])
    states = H.expand(ss, thetas, T)
    want = oracle.logl_batch(model.arrays(), model.measurement, c['err'], traj[:], states)
    got = model.logL_st_batch(ss, thetas, traj)
    assert got.shape == (n,)
    assert np.max(np.abs(got - want)) < TOL, (c['N'], c['d'], c['S'], T, c['nsw'], c['miss'], c['reduce'], c['path'])
    # expanded-profile entry point agrees bit for bit with the (s, theta) one
    assert np.array_equal(model.logL_batch(states, traj), got)
    # every evaluation above went through the trajectory set's tables (prefix table, convergence jumps, transient table)
    # where the kernel family has them; the frame-by-frame run of the same batch agrees to the jumps' tolerance
    if c['path'] == 'modal':
        exact = _lib.logl_st(model.handle(), model.trajset(traj), ss, thetas, path='modal', prefix=False)
        assert np.max(np.abs(exact - want)) < TOL
        assert np.max(np.abs(exact - got)) < 1e-9
