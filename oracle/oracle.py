"""
TEST INFRASTRUCTURE -- Python front-end of the CPU oracle (oracle/msrouse_logl.c) and
loaders for the reference's own kernels.

* `logl` / `logl_batch`      : the plain-C restatement (flavor 'cython' follows
                               bild/src/MSRouse_logL.pyx, flavor 'numpy' follows
                               bild/src/MSRouse_logL_py.py), called through ctypes.
* `load_reference_cython()`  : the reference's Cython kernel, compiled unmodified by
                               oracle/build_ref.py into oracle/_ref/ -- build container ONLY
                               (oracle/_ref/ is in .gpurunignore): validates the restatement
                               and yields oracle/conversion_factor.json; None elsewhere.
* `load_reference_numpy()`   : the reference's NumPy kernel, loaded by path from
                               /root/reference -- available in the build container only.

Parity status: pinned (tests/test_oracle.py checks the restatement against golden vectors
generated from both reference kernels, tests/golden/make_golden.py).
"""
import ctypes
import importlib.util
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLAVORS = {'numpy': 0, 'cython': 1}

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """ compile liboracle.so (gcc) next to this file """
    src = os.path.join(HERE, 'msrouse_logl.c')
    out = os.path.join(HERE, 'liboracle.so')
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-fPIC', '-std=c11', '-shared', src, '-o', out, '-lm'])
    return out


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.bild_oracle_logl.restype = ctypes.c_double
        _LIB.bild_oracle_logl.argtypes = ([ctypes.c_int] * 3 + [_dp] * 6
                                          + [ctypes.c_int, _dp, _ip, ctypes.c_int, _dp, _ip, ctypes.c_int])
        _LIB.bild_oracle_logl_batch.restype = None
        _LIB.bild_oracle_logl_batch.argtypes = ([ctypes.c_int] * 3 + [_dp] * 6
                                                + [ctypes.c_int, _dp, _ip, ctypes.c_int, _dp,
                                                   ctypes.c_int64, _ip, ctypes.c_int64, ctypes.c_int, _dp])
    return _LIB


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def noise_to_s2(localization_error):
    """ pyx:144-147: unique (sorted) errors squared, and the dim -> d* index """
    unique, Cind = np.unique(np.asarray(localization_error, dtype=np.float64), return_inverse=True)
    return unique * unique, Cind.astype(np.int32)


def logl_batch(arrays, w, localization_error, x, states, flavor='cython'):
    """
    Parameters
    ----------
    arrays : dict with B, G, Sig, M0, C0 stacked over states (bild_amd.rouse.stack_dynamics)
    w : (N,) measurement vector
    localization_error : (d,)
    x : (T, d) trajectory, NaN = missing
    states : (n, T) or (T,) int
    """
    B, G, Sig, M0, C0 = (_f64(arrays[k]) for k in ('B', 'G', 'Sig', 'M0', 'C0'))
    S, N, d = G.shape
    w = _f64(w)
    x = _f64(x)
    T = x.shape[0]
    assert x.shape == (T, d) and B.shape == (S, N, N) and w.shape == (N,)
    s2, Cind = noise_to_s2(localization_error)
    s2 = _f64(s2)
    states = np.ascontiguousarray(np.atleast_2d(states), dtype=np.int32)
    assert states.shape[1] == T
    out = np.empty(states.shape[0], dtype=np.float64)
    p = lambda a: a.ctypes.data_as(_dp)
    lib().bild_oracle_logl_batch(N, d, S, p(B), p(G), p(Sig), p(M0), p(C0), p(w),
                                 len(s2), p(s2), Cind.ctypes.data_as(_ip),
                                 T, p(x), states.shape[0], states.ctypes.data_as(_ip), T,
                                 FLAVORS[flavor], p(out))
    return out


def logl(arrays, w, localization_error, x, states, flavor='cython'):
    return float(logl_batch(arrays, w, localization_error, x, states, flavor)[0])


# ----------------------------------------------------------------------------------------
# the reference's own kernels
# ----------------------------------------------------------------------------------------
def load_reference_cython():
    """ MSRouse_logL(model, profile, traj) from the unmodified reference .pyx, or None """
    sys.path.insert(0, HERE)
    try:
        import build_ref
    finally:
        sys.path.pop(0)
    so = build_ref.build(verbose=False)
    if so is None:
        return None
    spec = importlib.util.spec_from_file_location('MSRouse_logL', so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.MSRouse_logL


def load_reference_numpy():
    """ MSRouse_logL from the reference's NumPy kernel (build container only), or None """
    path = '/root/reference/bild/src/MSRouse_logL_py.py'
    if not os.path.exists(path):
        return None
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only reference tree
    try:
        spec = importlib.util.spec_from_file_location('_bild_ref_MSRouse_logL_py', path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = old
    return mod.MSRouse_logL
