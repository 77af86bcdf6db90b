"""
CPU tests of the boundary: the C-ABI library loads without a GPU, exports every symbol
declared in include/bild_amd.h, and its host-side model analysis (invariant-subspace
reduction, modal decomposition) is mathematically consistent.  No compute calls here.
"""
import os
import re

import numpy as np
import pytest

import goldens
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(built_lib):
    header = open(os.path.join(ROOT, 'include', 'bild_amd.h')).read()
    declared = set(re.findall(r'\b(bild_[a-z_]+)\s*\(', header))
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(built_lib, name), f"{name} declared in include/bild_amd.h but not exported"
    from bild_amd import _lib
    assert declared == set(_lib.exported_symbols())
    assert built_lib.bild_abi_version() == 2


def test_loads_without_gpu_and_fails_loudly(built_lib):
    import bild_amd
    from bild_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    model = bild_amd.MultiStateRouse(20, 1, 5, d=1)
    traj = bild_amd.Trajectory([1, 2, np.nan, 4], localization_error=[0.5])
    with pytest.raises(_lib.NoDeviceError):      # no CPU fallback, ever
        model.logL(bild_amd.Loopingprofile([1, 1, 0, 0]), traj)


def test_no_localization_error_is_valueerror(built_lib):
    # reference bild/models.py:263, tested at tests/test_bild.py:140-143 -- raised before any device work
    import bild_amd
    model = bild_amd.MultiStateRouse(20, 1, 5, d=1)
    with pytest.raises(ValueError):
        model.logL(bild_amd.Loopingprofile([1, 1, 0, 0]), bild_amd.Trajectory([1, 2, np.nan, 4]))


def test_model_argument_validation(built_lib):
    from bild_amd import _lib
    a = H.DuckModel(N=6, d=2).arrays()
    w = H.end2end(6)
    with pytest.raises(AssertionError):
        _lib.ModelHandle(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], w[:-1])
    bad = a['B'].copy()
    bad[0, 0, 0] = np.nan
    with pytest.raises(_lib.BildAmdError):
        _lib.ModelHandle(bad, a['G'], a['Sig'], a['M0'], a['C0'], w)
    assert _lib.ModelHandle(a['B'], np.zeros((2, 6, 4)), a['Sig'], np.zeros((2, 6, 4)), a['C0'], w).query(_lib.Q_D) == 4
    with pytest.raises(_lib.BildAmdError):        # d = 9 is outside the compiled envelope (d <= 8)
        _lib.ModelHandle(a['B'], np.zeros((2, 6, 9)), a['Sig'], np.zeros((2, 6, 9)), a['C0'], w)


@pytest.mark.parametrize('N,loops,expect', [
    (20, H.LOOPS[2], 10),        # reflection-antisymmetric half
    (21, H.LOOPS[2], 10),
    (8, H.LOOPS[2], 4),
    (20, H.LOOPS[3], None),      # (0, 10) bond breaks the reflection symmetry: fewer modes decouple
    (12, (None, (1, 7)), None),
])
def test_invariant_subspace_reduction(built_lib, N, loops, expect):
    from bild_amd import _lib
    dm = H.DuckModel(N=N, loops=loops)
    a, w = dm.arrays(), dm.measurement
    h = _lib.ModelHandle(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], w)
    n = h.query(_lib.Q_NEFF)
    if expect is not None:
        assert n == expect
    assert 1 <= n <= N
    V = h.export(_lib.X_V)
    assert np.max(np.abs(V.T @ V - np.eye(n))) < 1e-13
    assert np.max(np.abs(V @ (V.T @ w) - w)) < 1e-13          # w lies in the subspace
    for key in ('B', 'Sig', 'C0'):
        for X in a[key]:
            assert np.max(np.abs(X @ V - V @ (V.T @ X @ V))) < 1e-12 * max(1., np.abs(X).max())  # invariant
    h_full = _lib.ModelHandle(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], w, reduce=False)
    assert h_full.query(_lib.Q_NEFF) == N


def _modal_filter_numpy(h, S, x, states, err):
    """ the modal algorithm the kernel implements, in NumPy, from the library's own host analysis """
    from bild_amd import _lib as L
    n = h.query(L.Q_NEFF)
    lam = [h.export(L.X_LAMBDA, s) for s in range(S)]
    sig = [h.export(L.X_SIGMA, s) for s in range(S)]
    wq = [h.export(L.X_WQ, s) for s in range(S)]
    C0q = [h.export(L.X_C0Q, s) for s in range(S)]
    R = {(a, b): h.export(L.X_R, a, b) for a in range(S) for b in range(S)}
    s = states[0]
    C, M, tot, s2 = C0q[s].copy(), np.zeros((n, x.shape[1])), 0., err ** 2
    for t in range(len(states)):
        if t > 0:
            sn = states[t]
            if sn != s:
                Rm = R[(s, sn)]
                C, M, s = Rm @ C @ Rm.T, Rm @ M, sn
            C = np.outer(lam[s], lam[s]) * C + np.diag(sig[s])
            M = lam[s][:, None] * M
        if not np.isnan(x[t]).any():
            Cw = C @ wq[s]
            Sv = wq[s] @ Cw + s2
            nu = x[t] - wq[s] @ M
            C = C - np.outer(Cw, Cw) / Sv
            M = M + np.outer(Cw, nu) / Sv
            tot += np.sum(-0.5 * (nu * nu / Sv + np.log(Sv) + np.log(2 * np.pi)))
    return tot


@pytest.mark.parametrize('name', ['s2_d3_T200', 's3_bursty_T300', 'asym_w_partial_nan_T60'])
@pytest.mark.parametrize('reduce', [True, False])
def test_host_analysis_reproduces_reference(built_lib, name, reduce):
    """ reduction + eigenbases exported by the library, run through a NumPy filter, hit the reference goldens """
    from bild_amd import _lib
    g = goldens.load(name)
    h = _lib.ModelHandle(g['B'], g['G'], g['Sig'], g['M0'], g['C0'], g['w'], reduce=reduce)
    assert h.query(_lib.Q_MODAL_OK) == 1
    S = g['B'].shape[0]
    for s in range(S):
        Q, lam, V = h.export(_lib.X_Q, s), h.export(_lib.X_LAMBDA, s), h.export(_lib.X_V)
        assert np.max(np.abs(Q.T @ Q - np.eye(len(lam)))) < 1e-13
        assert np.max(np.abs(Q @ np.diag(lam) @ Q.T - V.T @ g['B'][s] @ V)) < 1e-13
    err = g['localization_error']
    assert np.all(err == err[0])
    for i in range(min(4, len(g['states']))):
        got = _modal_filter_numpy(h, S, g['x'], g['states'][i], err[0])
        assert abs(got - g['logL_ref_cython'][i]) < 1e-8
        assert abs(got - g['logL_ref_numpy'][i]) < 1e-8


def test_modal_unavailable_for_nonsymmetric_propagator(built_lib):
    from bild_amd import _lib
    a = H.DuckModel(N=6, d=2).arrays()
    B = a['B'].copy()
    B[0, 0, 1] += 0.01      # the reference's two kernels already disagree on such input (dsymv vs full matmul)
    h = _lib.ModelHandle(B, a['G'], a['Sig'], a['M0'], a['C0'], H.end2end(6))
    assert h.query(_lib.Q_MODAL_OK) == 0 and h.query(_lib.Q_NEFF) == 6
    with pytest.raises(_lib.BildAmdError):
        h.export(_lib.X_Q, 0)


def test_collective_without_rccl_is_refused_not_crashed(tmp_path):
    """
    No RCCL to be found (the named library does not exist and nothing else is tried): the communicator calls return
    BILD_ERR_UNSUPPORTED with a message, as include/bild_amd.h promises -- they used to build that message from two
    dlerror() calls, the second of which returns NULL.  A fresh process: the library caches a loaded RCCL.
    """
    import subprocess
    import sys
    code = r'''
import ctypes, os, sys
sys.path.insert(0, sys.argv[1])
os.environ["BILD_AMD_RCCL_ONLY"] = "1"
os.environ["BILD_AMD_RCCL"] = os.path.join(sys.argv[2], "no_such_librccl.so")
from bild_amd import _lib
lib = _lib.lib()
lib.bild_comm_library(os.path.join(sys.argv[2], "neither.so").encode())
buf = ctypes.create_string_buffer(128)
rc = lib.bild_comm_unique_id(buf, 128)
msg = lib.bild_last_error().decode()
out = ctypes.c_void_p()
rc2 = lib.bild_comm_create(buf, 1, 0, ctypes.byref(out))
print(rc, rc2, msg)
sys.exit(0 if (rc == _lib.ERR_UNSUPPORTED and rc2 == _lib.ERR_UNSUPPORTED and "not found" in msg) else 1)
'''
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if os.environ.get('BILD_AMD_LIB'):
        env['BILD_AMD_LIB'] = os.environ['BILD_AMD_LIB']
    res = subprocess.run([sys.executable, '-c', code, root, str(tmp_path)], capture_output=True, text=True, env=env, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr


def test_config_is_read_once_and_reported(built_lib, monkeypatch):
    """ the BILD_* switches: read once into one struct, visible through bild_config_string, re-read only on request """
    from bild_amd import _lib
    _lib.config_reload()
    before = _lib.config_string()
    monkeypatch.setenv('BILD_NO_SPLIT', '1')
    monkeypatch.setenv('BILD_PAIRS_MAX_TASKS', '12345')
    assert _lib.config_string() == before                      # setting a variable changes nothing until a reload
    _lib.config_reload()
    now = _lib.config_string().split()
    assert 'BILD_NO_SPLIT=1' in now and 'BILD_PAIRS_MAX_TASKS=12345' in now
    monkeypatch.delenv('BILD_NO_SPLIT')
    monkeypatch.delenv('BILD_PAIRS_MAX_TASKS')
    _lib.config_reload()
    assert _lib.config_string() == before


def test_negative_cumulative_position_is_refused(built_lib):
    """
    bild/amis.py:685-688 takes np.floor of cumsum(s) * (T - 1): a position in (-1, 0) gives switch index 0, i.e. a profile
    whose FIRST interval is empty.  A truncating conversion would give index 1 instead -- silently another profile --, so
    such rows are refused on every path (include/bild_amd.h: bild_logl_st), as the header says.
    """
    from bild_amd import _lib
    ss = np.array([[-1e-9, 0.5, 0.5 + 1e-9]])
    thetas = np.array([[0, 1, 0]])
    T = 100
    assert (np.floor(np.cumsum(ss[0])[:-1] * (T - 1)).astype(int) + 1).tolist() == [0, 50]     # what the reference computes
    with pytest.raises(_lib.BildAmdError, match="simplex"):
        _lib.segments_from_st(ss, thetas, T, 2)
    ok = np.array([[0.0, 0.5, 0.5], [1e-300, 0.5, 0.5]])       # zero and tiny positive first intervals are fine
    a, _ = _lib.segments_from_st(ok, np.array([[0, 1, 0]] * 2), T, 2)
    assert a.tolist() == [[0, 1, 50], [0, 1, 50]]
    # -0.0 is not negative: np.floor(-0.0) + 1 = 1, and so here
    a, _ = _lib.segments_from_st(np.array([[-0.0, 0.5, 0.5]]), thetas, T, 2)
    assert a.tolist() == [[0, 1, 50]]
