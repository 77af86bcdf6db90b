"""
CPU tests: the oracle (oracle/msrouse_logl.c, a plain-C restatement of the reference
algorithm) is pinned against golden vectors produced by the reference's own NumPy and Cython
kernels (tests/golden/make_golden.py).  Tolerance: |delta| < 1e-8 absolute on logL, the
bar BASELINE.json states; observed agreement is ~1e-11.
"""
import numpy as np
import pytest

import goldens
from oracle import oracle

TOL = 1e-8


@pytest.mark.parametrize('name', goldens.names())
@pytest.mark.parametrize('flavor', ['cython', 'numpy'])
def test_oracle_matches_reference_goldens(name, flavor):
    g = goldens.load(name)
    got = oracle.logl_batch(g['arrays'], g['w'], g['localization_error'], g['x'], g['states'], flavor=flavor)
    for key in ('logL_ref_numpy', 'logL_ref_cython'):
        ref = g[key]
        ok = ~np.isnan(ref)
        assert np.all(np.isfinite(got))
        if np.any(ok):
            assert np.max(np.abs(got[ok] - ref[ok])) < TOL, (name, flavor, key)


def test_reference_unittest_fixture_range():
    # reference tests/test_bild.py:138 pins -100 < logL < 0 for its 4-frame fixture
    g = goldens.load('ref_unittest_4frames')
    got = oracle.logl_batch(g['arrays'], g['w'], g['localization_error'], g['x'], g['states'][:1])
    assert -100 < got[0] < 0
    # ... and Cython == NumPy on it (tests/test_bild.py:168-173)
    assert abs(g['logL_ref_cython'][0] - g['logL_ref_numpy'][0]) < 1e-12


def test_all_missing_is_zero():
    g = goldens.load('all_missing_T10')
    got = oracle.logl_batch(g['arrays'], g['w'], g['localization_error'], g['x'], g['states'])
    assert np.all(got == 0.0) and np.all(g['logL_ref_numpy'] == 0.0)


def test_prebuilt_reference_cython_matches_goldens():
    """ the unmodified reference kernel, built into oracle/_ref in the build container (it does not travel: skipped elsewhere) """
    import helpers as H
    ref = oracle.load_reference_cython()
    if ref is None:
        pytest.skip("oracle/_ref not built and /root/reference absent")
    g = goldens.load('s2_d3_T200')

    class M:
        d = g['x'].shape[1]
        measurement = g['w']

        def _get_noise(self, traj):
            return g['localization_error']
    m = M()
    m.models = H.DuckModel(N=20, D=1, k=5, d=3).models  # same builder as the fixture
    a = H.rouse.stack_dynamics(m.models)
    assert np.array_equal(a['B'], g['B'])  # builder is deterministic: fixture inputs are reproducible
    for i in (0, 5, 16):
        got = ref(m, H.ProfileView(g['states'][i]), g['x'])
        assert abs(got - g['logL_ref_cython'][i]) < 1e-12
