"""
The file a BILD maintainer would add to the reference tree as ``bild/gpu_imports.py`` (see
INTEGRATION.md): a ctypes binding of the C ABI in include/bild_amd.h with the exact signature of the
reference's native kernel, ``MSRouse_logL(model, profile, traj) -> float``
(bild/src/MSRouse_logL.pyx:95-98), plus the batch hook for ``FixedkSampler.logL``
(bild/amis.py:734-739).  It reads the same attributes of ``model`` / ``traj`` the Cython kernel
reads (pyx:144-178) and nothing else, and depends only on NumPy and the shared library.

Set ``BILD_AMD_LIB`` to the path of ``libbild_amd.so`` if it is not on the loader path.
"""
import ctypes
import os

import numpy as np

_lib = ctypes.CDLL(os.environ.get('BILD_AMD_LIB', 'libbild_amd.so'))
_dp, _ip, _vp = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p
_lib.bild_last_error.restype = ctypes.c_char_p
_lib.bild_model_create.argtypes = [ctypes.c_int] * 3 + [_dp] * 6 + [ctypes.c_uint, ctypes.POINTER(_vp)]
_lib.bild_trajset_create.argtypes = [_vp, ctypes.c_int, _ip, _dp, _dp, ctypes.POINTER(_vp)]
_lib.bild_logl_profiles.argtypes = [_vp, _vp, ctypes.c_int64, ctypes.c_int64, _ip, _ip, ctypes.c_uint, _dp]
_lib.bild_logl_segments.argtypes = [_vp, _vp, ctypes.c_int64, ctypes.c_int, _ip, _ip, _ip, ctypes.c_uint, _dp]

_models, _trajs = {}, {}


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def _check(rc):
    if rc:
        raise RuntimeError(_lib.bild_last_error().decode())


def _model_handle(model):
    h = _models.get(id(model))
    if h is None:
        for m in model.models:
            m.check_dynamics()                                                        # pyx:152-153
        B, G, Sig = (_f64([m._dynamics[k] for m in model.models]) for k in ('B', 'G', 'Sig'))   # pyx:155-157
        ss = [m.steady_state() for m in model.models]                                 # pyx:160
        M0, C0 = _f64([s[0] for s in ss]), _f64([s[1] for s in ss])
        w = _f64(model.measurement)                                                   # pyx:150
        S, N, d = G.shape
        h = _vp()
        _check(_lib.bild_model_create(N, d, S, _p(B), _p(G), _p(Sig), _p(M0), _p(C0), _p(w), 0, ctypes.byref(h)))
        _models[id(model)] = h
    return h


def _traj_handle(model, traj):
    mh = _model_handle(model)
    key = (id(model), id(traj))
    th = _trajs.get(key)
    if th is None:
        x = _f64(traj[:])                                                             # pyx:174
        if x.ndim == 1:
            x = x[:, None]
        err = _f64(model._get_noise(traj))                                            # pyx:144
        T = np.array([len(x)], dtype=np.int32)
        th = _vp()
        _check(_lib.bild_trajset_create(mh, 1, T.ctypes.data_as(_ip), _p(x), _p(err), ctypes.byref(th)))
        _trajs[key] = th
    return mh, th


def MSRouse_logL(model, profile, traj):
    """ drop-in for bild.cython_imports.MSRouse_logL """
    mh, th = _traj_handle(model, traj)
    states = np.ascontiguousarray(profile[:], dtype=np.int32)                         # pyx:175
    out = np.empty(1)
    _check(_lib.bild_logl_profiles(mh, th, 1, len(states), states.ctypes.data_as(_ip), None, 0, _p(out)))
    return float(out[0])


def logL_st_batch(model, ss, thetas, traj):
    """ what FixedkSampler.logL hands over when the model offers ``logL_st_batch`` (one launch per AMIS step) """
    mh, th = _traj_handle(model, traj)
    T = len(traj)
    thetas = np.asarray(thetas)
    seg_start = np.zeros(thetas.shape, dtype=np.int32)
    if thetas.shape[1] > 1:                                                           # amis.py:685-688
        seg_start[:, 1:] = np.floor(np.cumsum(np.asarray(ss, dtype=float), axis=1)[:, :-1] * (T - 1)).astype(int) + 1
    seg_state = np.ascontiguousarray(thetas, dtype=np.int32)
    out = np.empty(len(thetas))
    _check(_lib.bild_logl_segments(mh, th, len(thetas), thetas.shape[1], seg_start.ctypes.data_as(_ip),
                                   seg_state.ctypes.data_as(_ip), None, 0, _p(out)))
    return out
