"""
GPU test of the reference-side binding shown in INTEGRATION.md (integration/gpu_imports.py): a plain
ctypes stub with the reference kernel's signature ``MSRouse_logL(model, profile, traj)``, fed the same
duck-typed objects the reference kernels are fed, must reproduce the reference goldens.
"""
import importlib.util
import os

import numpy as np
import pytest

import goldens
import helpers as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def stub(built_lib):
    os.environ['BILD_AMD_LIB'] = os.path.join(ROOT, 'bild_amd', 'libbild_amd.so')
    spec = importlib.util.spec_from_file_location('gpu_imports', os.path.join(ROOT, 'integration', 'gpu_imports.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_stub_reproduces_reference_goldens(stub):
    # same duck-typed model builder as the golden generator (deterministic)
    g = goldens.load('s2_d3_T200')
    model = H.DuckModel(N=20, D=1, k=5, d=3, localization_error=0.1)
    assert np.array_equal(model.arrays()['B'], g['B'])
    from bild_amd.trajectory import Trajectory
    traj = Trajectory(g['x'], localization_error=g['localization_error'])
    for i in (0, 3, 9, 16):
        got = stub.MSRouse_logL(model, H.ProfileView(g['states'][i]), traj)
        assert isinstance(got, float)
        assert abs(got - g['logL_ref_cython'][i]) < 1e-8 and abs(got - g['logL_ref_numpy'][i]) < 1e-8

    g = goldens.load('ref_unittest_4frames')          # the reference's own unit-test fixture, d = 1
    model = H.DuckModel(N=20, D=1, k=5, d=1)
    traj = Trajectory([1, 2, np.nan, 4], localization_error=[0.5])
    got = stub.MSRouse_logL(model, H.ProfileView([1, 1, 0, 0]), traj)
    assert -100 < got < 0 and abs(got - g['logL_ref_cython'][0]) < 1e-10


def test_stub_batch_hook_matches_per_profile_calls(stub):
    rng = np.random.default_rng(2)
    model = H.DuckModel(N=20, D=1, k=5, d=3, localization_error=[0.1, 0.1, 0.3])
    T = 90
    traj = H.synth_trajectory(model, H.random_profile(rng, T, 2, 30), [0.1, 0.1, 0.3], rng, missing=[0, 40, 41])
    ss, thetas = H.candidate_profiles(rng, 30, 3, 2)
    batch = stub.logL_st_batch(model, ss, thetas, traj)
    states = H.expand(ss, thetas, T)
    single = np.array([stub.MSRouse_logL(model, H.ProfileView(st), traj) for st in states])
    assert np.array_equal(batch, single)
