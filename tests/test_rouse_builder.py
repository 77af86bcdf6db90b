"""
CPU tests of the Rouse matrix builder (bild_amd/rouse.py): internal consistency of the exact
one-frame propagator with the steady state, and -- only where the third-party `rouse` package is
importable (it is not in the build container: "Level M" parity is unpinned) -- a direct
comparison with ``rouse.Model``.
"""
import numpy as np
import pytest

from bild_amd import rouse as own


@pytest.mark.parametrize('N,bonds', [(5, None), (20, None), (20, [(0, -1)]), (12, [(1, 7, 2.0)]), (9, [(0, 4), (4, 8)])])
def test_propagator_and_steady_state_are_consistent(N, bonds):
    m = own.Model(N, D=0.7, k=3.0, d=3, add_bonds=bonds)
    m.check_dynamics()
    B, Sig, C0 = m._dynamics['B'], m._dynamics['Sig'], m._dynamics['C0']
    assert np.allclose(B, B.T, atol=1e-14) and np.allclose(Sig, Sig.T, atol=1e-14)
    # the steady state is a fixed point of the covariance propagation on the internal modes ...
    assert np.allclose(B @ C0 @ B + Sig - C0, np.full_like(C0, (B @ C0 @ B + Sig - C0)[0, 0]), atol=1e-11)
    # ... and the only non-stationary direction is the free centre-of-mass mode (uniform vector), which
    # gains 2 D per frame spread over the N monomers
    drift = B @ C0 @ B + Sig - C0
    assert np.allclose(drift, 2 * 0.7 / N, atol=1e-11)
    # uniform vector: eigenvector of B with eigenvalue 1 (all rows of the Laplacian sum to zero)
    assert np.allclose(B @ np.ones(N), np.ones(N), atol=1e-12)
    # composition: two frames == one frame of twice the rate constant only if k scales time; check the
    # semigroup property instead: B(k) B(k) == B(2k)
    m2 = own.Model(N, D=0.7, k=6.0, d=3, add_bonds=bonds)
    m2.check_dynamics()
    assert np.allclose(B @ B, m2._dynamics['B'], atol=1e-12)


def test_external_force_shifts_the_mean():
    m = own.Model(8, D=1, k=2, d=2)
    m.F[0] = [1.0, -0.5]
    m.F[-1] = [-1.0, 0.5]
    m.update_dynamics()
    M0, _ = m.steady_state()
    assert np.allclose(m.propagate_M(M0), M0, atol=1e-12)          # fixed point of the mean propagation
    assert np.abs(M0).max() > 0.1


def test_matches_rouse_package_if_installed():
    rouse = pytest.importorskip('rouse')
    for N, bonds in [(20, None), (20, [(0, -1)])]:
        ref = rouse.Model(N, 1., 5., 3, add_bonds=bonds)
        ref.check_dynamics()
        mine = own.Model(N, 1., 5., 3, add_bonds=bonds)
        mine.check_dynamics()
        for key in ('B', 'G', 'Sig'):
            assert np.allclose(mine._dynamics[key], ref._dynamics[key], atol=1e-12), key
        # the steady state may differ in the centre-of-mass convention only: compare on sum(w) = 0 observables
        w = np.zeros(N)
        w[0], w[-1] = -1, 1
        assert abs(w @ mine.steady_state()[1] @ w - w @ ref.steady_state()[1] @ w) < 1e-10


def test_initial_loopingprofile_pins_of_the_reference():
    """ reference tests/test_bild.py:129-133 (base implementation) and :145-151 (Rouse -> factorized guess) """
    import bild_amd
    from bild_amd.models import MultiStateModel
    traj = bild_amd.Trajectory([1, 2, np.nan, 4], localization_error=[0.5])
    model = bild_amd.MultiStateRouse(20, 1, 5, d=1)
    assert len(MultiStateModel.initial_loopingprofile(model, traj)) == 4                  # :132-133
    model = bild_amd.MultiStateRouse(20, 1, 5, d=1, localization_error=0.5)
    assert np.array_equal(model.initial_loopingprofile(traj).state, [1, 0, 0, 0])         # :150-151
    fact = model.toFactorized()                                                           # models.py:352-370
    assert fact.nStates == 2 and fact.d == 1
    w = model.measurement
    for dist, C0 in zip(fact.distributions, model.arrays()['C0']):
        assert np.isclose(dist.kwds['scale'] ** 2, w @ C0 @ w + 0.25)


def test_factorized_model_pins_of_the_reference():
    """ reference tests/test_bild.py:175-195 """
    import bild_amd
    from scipy import stats
    traj = bild_amd.Trajectory([1, 2, np.nan, 4], localization_error=[0.5])
    profile = bild_amd.Loopingprofile([1, 1, 0, 0])
    model = bild_amd.FactorizedModel([stats.maxwell(scale=1), stats.maxwell(scale=4)], d=1)
    assert model.nStates == 2
    for _ in range(2):
        assert -100 < model.logL(profile, traj) < 0
        assert np.array_equal(model.initial_loopingprofile(traj).state, [0, 0, 1, 1])
        model.clear_memo()
    gen = model.trajectory_from_loopingprofile(bild_amd.Loopingprofile([0, 0, 0, 1, 1, 1]))
    assert len(gen) == 6 and gen[:].shape == (6, 1)
    gen = model.trajectory_from_loopingprofile(bild_amd.Loopingprofile([0, 0, 0, 1, 1, 1]), missing_frames=[1, 4])
    assert np.array_equal(np.isnan(gen[:][:, 0]), [False, True, False, False, True, False])


def test_from_reference_takes_the_matrices_as_they_are():
    """ `MultiStateRouse.from_reference` on the attribute surface the reference kernel reads (pyx:150-160) """
    import bild_amd
    import helpers as H
    duck = H.DuckModel(N=12, D=1., k=3., d=2, localization_error=[0.1, 0.2])
    model = bild_amd.MultiStateRouse.from_reference(duck)
    want = duck.arrays()
    got = model.arrays()
    for key in ('B', 'G', 'Sig', 'M0', 'C0'):
        assert np.array_equal(np.asarray(got[key]), np.asarray(want[key])), key
    assert np.array_equal(model.measurement, duck.measurement) and model.nStates == 2 and model.d == 2
    assert np.array_equal(model.localization_error, [0.1, 0.2])
    assert np.array_equal(model.initial_loopingprofile(bild_amd.Trajectory(np.array([[0.1, 0.1], [3., 2.]])))[:].shape, (2,))
