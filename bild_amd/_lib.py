"""
ctypes binding of ``libbild_amd.so`` (C ABI: include/bild_amd.h).

There is no CPU fallback: if the shared library is missing or no GPU is usable, every
evaluation raises.  (The reference's warn-and-fall-back shim, bild/cython_imports.py:3-7,
is deliberately *not* mirrored: a silent fallback would void the parity claims.)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BILD_AMD_LIB selects an alternative build of the same ABI (kernel A/B experiments, tools/ab.py)
LIB_PATH = os.environ.get('BILD_AMD_LIB') or os.path.join(_HERE, 'libbild_amd.so')

OK, ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_NOMEM = range(6)

PATH_AUTO, PATH_DENSE, PATH_MODAL = 0, 1, 2
PATHS = {'auto': PATH_AUTO, 'dense': PATH_DENSE, 'modal': PATH_MODAL}
MODEL_NO_REDUCE = 1
VALIDATE_DEVICE = 0x10
NO_PREFIX = 0x20
NO_JUMP = 0x40
NO_SPLIT = 0x80
NO_STATES = 0x100
NO_TAIL = 0x200

Q_N, Q_D, Q_S, Q_MODAL_OK, Q_NP, Q_NEFF, Q_HAS_G = range(7)
X_LAMBDA, X_SIGMA, X_Q, X_WQ, X_R, X_C0Q, X_V = range(7)


class BildAmdError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"bild_amd error {code}: {message}")
        self.code = code


class NoDeviceError(BildAmdError):
    pass


# (double* / int32* parameters are declared as addresses: a NumPy buffer is then passed as a plain integer -- dptr / iptr below --,
# which costs a third of building a typed ctypes pointer per argument; byref(...) of an output scalar is accepted as before)
_dp = ctypes.c_void_p
_ip = ctypes.c_void_p
_vp = ctypes.c_void_p

_SIGNATURES = {
    'bild_abi_version': (ctypes.c_int, []),
    'bild_config_string': (ctypes.c_char_p, []),
    'bild_config_reload': (ctypes.c_int, []),
    'bild_last_error': (ctypes.c_char_p, []),
    'bild_set_last_error': (None, [ctypes.c_char_p]),
    'bild_comm_library': (ctypes.c_int, [ctypes.c_char_p]),
    'bild_comm_unique_id': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    'bild_comm_create': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    'bild_comm_allgather': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, _vp]),
    'bild_comm_destroy': (ctypes.c_int, [_vp]),
    'bild_device_alloc': (ctypes.c_int, [ctypes.c_int64, ctypes.POINTER(_vp)]),
    'bild_device_free': (ctypes.c_int, [_vp]),
    'bild_device_to_host': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _vp]),
    'bild_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'bild_model_create': (ctypes.c_int, [ctypes.c_int] * 3 + [_dp] * 6 + [ctypes.c_uint, ctypes.POINTER(_vp)]),
    'bild_model_destroy': (ctypes.c_int, [_vp]),
    'bild_model_query': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
    'bild_model_export': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, ctypes.c_int64]),
    'bild_trajset_create': (ctypes.c_int, [_vp, ctypes.c_int, _ip, _dp, _dp, ctypes.POINTER(_vp)]),
    'bild_trajset_destroy': (ctypes.c_int, [_vp]),
    'bild_trajset_expect': (ctypes.c_int, [_vp, ctypes.c_int64]),
    'bild_logl_segments': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, _ip, _ip, _ip, ctypes.c_uint, _dp]),
    'bild_logl_st': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, _dp, _vp, _ip, ctypes.c_uint, _dp]),
    'bild_logl_st_to_device': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, _dp, _vp, _ip, ctypes.c_uint, _vp, _vp]),
    'bild_logl_st_device': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, _vp, _vp, _vp, ctypes.c_uint, _vp, _vp, _vp]),
    'bild_segments_from_st': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int64, _dp, _vp, _ip, _ip]),
    'bild_logl_profiles': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int64, _ip, _ip, ctypes.c_uint, _dp]),
    'bild_logl_segments_device': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, _vp, _vp, _vp,
                                                 ctypes.c_uint, _vp, _vp]),
    'bild_schedule_segments': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, _ip, _ip, _ip, ctypes.c_uint, _ip]),
    'bild_logl_segments_device_ordered': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, _vp, _vp, _vp, _vp,
                                                         ctypes.c_uint, _vp, _vp]),
    'bild_frames_executed': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, ctypes.c_int, _ip, _ip, _ip, ctypes.c_uint, _dp, _dp]),
    'bild_frames_run_read': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    'bild_debug_frames_per_task': (ctypes.c_int, [_vp]),
    'bild_prefix_info': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64), _dp]),
    'bild_flop_count': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _ip, ctypes.c_uint, _dp, _dp]),
    'bild_kernel_timing': (ctypes.c_int, [ctypes.c_int]),
    'bild_kernel_timing_read': (ctypes.c_int, [_dp, ctypes.POINTER(ctypes.c_int64), ctypes.c_char_p, ctypes.c_int]),
    'bild_kernel_timing_read_walk': (ctypes.c_int, [_dp, ctypes.POINTER(ctypes.c_int64)]),
    'bild_logl_st_status': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64)]),
    # host-side AMIS bookkeeping (amis_host.cpp)
    'bild_amis_create': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                        _dp, _dp, ctypes.POINTER(_vp)]),
    'bild_amis_destroy': (ctypes.c_int, [_vp]),
    'bild_amis_error': (ctypes.c_char_p, [_vp]),
    'bild_amis_pool_size': (ctypes.c_int64, [_vp]),
    'bild_amis_num_proposals': (ctypes.c_int64, [_vp]),
    'bild_amis_params': (ctypes.c_int, [_vp, ctypes.c_int64, _dp, _dp]),
    'bild_amis_pool': (ctypes.c_int, [_vp, ctypes.c_int, _dp]),
    'bild_amis_restore': (ctypes.c_int, [_vp, ctypes.c_int64, _dp, _dp, ctypes.c_int64, _dp, _vp, _dp, _dp, _dp, _dp]),
    'bild_amis_sample_traces': (ctypes.c_int, [_vp, ctypes.c_int64, _dp, _vp]),
    'bild_amis_step': (ctypes.c_int, [_vp, ctypes.c_int64, _dp, _vp, _dp, _dp]),
    'bild_amis_use_device': (ctypes.c_int, [_vp, ctypes.c_int]),
    'bild_amis_step_fused': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, _dp, _vp, ctypes.c_uint, _dp]),
    'bild_amis_step_device_rng': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint, _dp]),
    'bild_amis_pool_samples': (ctypes.c_int, [_vp, _dp, _vp]),
    'bild_interval_marginals': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int64, _ip, _ip, _dp, _dp]),
    'bild_exchange_create': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.POINTER(_vp)]),
    'bild_exchange_handle': (ctypes.c_int, [_vp, ctypes.c_char_p]),
    'bild_exchange_connect': (ctypes.c_int, [_vp, ctypes.c_char_p]),
    'bild_exchange_allgather': (ctypes.c_int, [_vp, _vp, ctypes.c_int64, _vp, _vp]),
    'bild_exchange_status': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int)]),
    'bild_exchange_set_step': (ctypes.c_int, [_vp, ctypes.c_uint32]),
    'bild_exchange_set_timeout': (ctypes.c_int, [_vp, ctypes.c_double]),
    'bild_exchange_destroy': (ctypes.c_int, [_vp]),
    # the inference driver (run_host.cpp)
    'bild_run_create': (ctypes.c_int, [ctypes.c_int, _ip, ctypes.c_int, _vp, _vp, ctypes.c_int, _dp, _dp, _dp, _vp, _ip,
                                       ctypes.POINTER(_vp)]),
    'bild_run_destroy': (ctypes.c_int, [_vp]),
    'bild_run_error': (ctypes.c_char_p, [_vp]),
    'bild_run_plan': (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_vp)]),
    'bild_run_round': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint, _dp, _dp, _dp]),
    'bild_run_stage': (ctypes.c_int, [_vp, _dp, _dp]),
    'bild_run_rows': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_vp),
                                     ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    'bild_run_finish': (ctypes.c_int, [_vp, _dp, _dp]),
    'bild_run_traj_info': (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.POINTER(ctypes.c_char_p)]),
    'bild_run_traj_log': (ctypes.c_int, [_vp, ctypes.c_int, _ip, _ip, _dp, _dp, _dp]),
    'bild_run_sampler_info': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp]),
    'bild_run_sampler_data': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _dp, _dp, _vp, _dp]),
    'bild_run_take_core': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    'bild_run_totals': (ctypes.c_int, [_vp, _vp]),
    'bild_choice_counts': (ctypes.c_int, [ctypes.c_int64, ctypes.c_int, _dp, _dp, _dp, ctypes.c_double, _vp, _vp, _vp, _vp]),
}

_lib = None
_torch_libdir = None     # set when the library was bound to the HIP runtime of an installed PyTorch
_rccl_chosen = False


def _share_hip_runtime_with_torch():
    """
    A PyTorch-ROCm wheel ships its own HIP runtime (same soname as the system one).  Two runtimes in one process do
    not coexist: if this library brings in the system runtime first, a later ``torch.cuda`` finds "No HIP GPUs".
    So when PyTorch is installed but not loaded yet, its runtime is loaded first and the library binds to it -- the
    same arrangement as when torch is imported before bild_amd.  ``BILD_AMD_HIP_RUNTIME=system`` skips this.
    """
    import importlib.util
    import sys
    if 'torch' in sys.modules or os.environ.get('BILD_AMD_HIP_RUNTIME') == 'system':
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            global _torch_libdir
            _torch_libdir = os.path.dirname(path)
        except OSError:
            pass


def lib():
    """ load libbild_amd.so (raises if it has not been built) """
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"or `make -C bild_amd/csrc` (there is no CPU fallback)")
        _share_hip_runtime_with_torch()
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.bild_abi_version() != 2:
            raise ImportError("libbild_amd.so ABI version mismatch")
        _lib = handle
    return _lib


def config_string():
    """ the BILD_* environment switches in force, as the library read them ("" = defaults) """
    return lib().bild_config_string().decode()


def config_reload():
    """ re-read the BILD_* environment switches (tests / tools; not while evaluations are running) """
    check(lib().bild_config_reload())


def exported_symbols():
    return list(_SIGNATURES)


def check(code):
    if code != OK:
        msg = lib().bild_last_error().decode()
        raise (NoDeviceError if code == ERR_NO_DEVICE else BildAmdError)(code, msg)


def device_count():
    c = ctypes.c_int(0)
    check(lib().bild_device_count(ctypes.byref(c)))
    return c.value


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


_addressof, _from_buffer = ctypes.addressof, ctypes.c_char.from_buffer


def aptr(a):
    """
    address of a C-contiguous array of any element type; the CALLER keeps the array alive across the call (no temporaries
    here).  Through the buffer protocol where that works (0.4 us; writable, non-empty arrays), else through
    ``__array_interface__`` (1.2 us; building a typed ctypes pointer costs 2.3).
    """
    try:
        return _addressof(_from_buffer(a))
    except (TypeError, ValueError, BufferError):
        return a.__array_interface__['data'][0]


dptr = aptr   # float64 buffers


def iptr(a):
    """ address of a C-contiguous int32 array, or None; lifetime as for `aptr` """
    return None if a is None else aptr(a)


class ModelHandle:
    """ owns a ``bild_model*`` """

    def __init__(self, B, G, Sig, M0, C0, w, reduce=True):
        B, G, Sig, M0, C0, w = (f64(a) for a in (B, G, Sig, M0, C0, w))
        S, N, d = G.shape
        if B.shape != (S, N, N) or Sig.shape != (S, N, N) or C0.shape != (S, N, N) \
                or M0.shape != (S, N, d) or w.shape != (N,):
            raise AssertionError("inconsistent model array shapes")  # reference: pyx:165-166 asserts
        self.N, self.d, self.S = N, d, S
        self._h = _vp()
        check(lib().bild_model_create(N, d, S, dptr(B), dptr(G), dptr(Sig), dptr(M0), dptr(C0), dptr(w),
                                      0 if reduce else MODEL_NO_REDUCE, ctypes.byref(self._h)))

    def query(self, what):
        v = ctypes.c_int64(0)
        check(lib().bild_model_query(self._h, what, ctypes.byref(v)))
        return v.value

    def export(self, what, s=0, s2=0):
        n = self.query(Q_NEFF)
        shape = {X_LAMBDA: (n,), X_SIGMA: (n,), X_Q: (n, n), X_WQ: (n,), X_R: (n, n), X_C0Q: (n, n),
                 X_V: (self.N, n)}[what]
        buf = np.empty(shape, dtype=np.float64)
        check(lib().bild_model_export(self._h, what, s, s2, dptr(buf), buf.size))
        return buf

    def __del__(self):
        if getattr(self, '_h', None) and _lib is not None:
            _lib.bild_model_destroy(self._h)
            self._h = None


class TrajSetHandle:
    """ owns a ``bild_trajset*`` (device-resident trajectories) """

    def __init__(self, model, trajs, loc_errs):
        self.model = model  # keep alive
        arrs = [f64(t) for t in trajs]
        for a in arrs:
            if a.ndim != 2 or a.shape[1] != model.d:
                raise AssertionError(f"trajectory shape {a.shape} does not match model dimension d={model.d}")
        self.T = i32([a.shape[0] for a in arrs])
        x = f64(np.concatenate(arrs, axis=0))
        err = f64(loc_errs).reshape(len(arrs), model.d)
        self.n_traj = len(arrs)
        self._h = _vp()
        check(lib().bild_trajset_create(model._h, self.n_traj, iptr(self.T), dptr(x), dptr(err), ctypes.byref(self._h)))

    def expect(self, evaluations):
        """ declare, before the first evaluation, how many evaluations the set will see (bild_trajset_expect) """
        check(lib().bild_trajset_expect(self._h, int(evaluations)))

    def __del__(self):
        if getattr(self, '_h', None) and _lib is not None:
            _lib.bild_trajset_destroy(self._h)
            self._h = None


def logl_segments(model, ts, seg_start, seg_state, traj_id=None, path='auto', prefix=True, jump=True, split=True, states=True, tail=True):
    seg_start, seg_state = i32(seg_start), i32(seg_state)
    n, K1 = seg_start.shape
    assert seg_state.shape == (n, K1)
    tid = None if traj_id is None else i32(traj_id)
    out = np.empty(n, dtype=np.float64)
    check(lib().bild_logl_segments(model._h, ts._h, n, K1, iptr(seg_start), iptr(seg_state), iptr(tid),
                                   _flags(path, prefix=prefix, jump=jump, split=split, states=states, tail=tail), dptr(out)))
    return out


def logl_st(model, ts, ss, thetas, traj_id=None, path='auto', prefix=True, jump=True, split=True, tail=True):
    """ the sampler's (s, theta) batch as it is: switch frames are computed natively (bild_logl_st) """
    ss = f64(ss)
    thetas = np.ascontiguousarray(thetas, dtype=np.int64)
    if ss.ndim == 1:
        ss, thetas = ss[None, :], thetas[None, :]
    n, K1 = thetas.shape
    assert ss.shape == (n, K1)
    tid = None if traj_id is None else i32(traj_id)
    out = np.empty(n, dtype=np.float64)
    check(lib().bild_logl_st(model._h, ts._h, n, K1, dptr(ss), aptr(thetas), iptr(tid), _flags(path, prefix=prefix, jump=jump, split=split, tail=tail), dptr(out)))
    return out


def logl_st_to_device(model, ts, ss, thetas, d_out, traj_id=None, stream=0, path='auto'):
    """ host (s, theta) batch in, results left in HBM at the raw device pointer `d_out`; asynchronous on `stream` """
    ss = f64(ss)
    thetas = np.ascontiguousarray(thetas, dtype=np.int64)
    n, K1 = thetas.shape
    assert ss.shape == (n, K1)
    tid = None if traj_id is None else i32(traj_id)
    check(lib().bild_logl_st_to_device(model._h, ts._h, n, K1, dptr(ss), aptr(thetas), iptr(tid), PATHS[path],
                                       _vp(stream) if stream else None, _vp(d_out)))


def segments_from_st(ss, thetas, T, n_states):
    """ native (s, theta) -> run-length segments (bild_segments_from_st); T: one length or one per sample """
    ss = f64(ss)
    thetas = np.ascontiguousarray(thetas, dtype=np.int64)
    n, K1 = thetas.shape
    assert ss.shape == (n, K1)
    Ts = i32(np.atleast_1d(T))
    assert len(Ts) in (1, n)
    seg_start = np.empty((n, K1), dtype=np.int32)
    seg_state = np.empty((n, K1), dtype=np.int32)
    check(lib().bild_segments_from_st(n, K1, int(n_states), iptr(Ts), 0 if len(Ts) == 1 else 1, dptr(ss),
                                      aptr(thetas), iptr(seg_start), iptr(seg_state)))
    return seg_start, seg_state


def logl_profiles(model, ts, states, traj_id=None, path='auto'):
    states = i32(np.atleast_2d(states))
    n, ld = states.shape
    tid = None if traj_id is None else i32(traj_id)
    out = np.empty(n, dtype=np.float64)
    check(lib().bild_logl_profiles(model._h, ts._h, n, ld, iptr(states), iptr(tid), PATHS[path], dptr(out)))
    return out


def _flags(path, validate=False, prefix=True, jump=True, split=True, states=True, tail=True):
    return (PATHS[path] | (VALIDATE_DEVICE if validate else 0) | (0 if prefix else NO_PREFIX) | (0 if jump else NO_JUMP) |
            (0 if split else NO_SPLIT) | (0 if states else NO_STATES) | (0 if tail else NO_TAIL))


def frames_run_read(model):
    """ frames the tasks ran themselves since the last call (kernel timing must be on); resets the counter """
    v = ctypes.c_int64(0)
    check(lib().bild_frames_run_read(model._h, ctypes.byref(v)))
    return v.value


def schedule_segments(model, ts, seg_start, seg_state, traj_id=None, path='auto', prefix=True, jump=True):
    """
    launch order for device-resident candidates (bild_schedule_segments): (n,) int32, order[slot] = sample.  The work
    estimate behind it uses the trajectory set's tables: evaluate something on the set once before asking.
    """
    seg_start, seg_state = i32(seg_start), i32(seg_state)
    n, K1 = seg_start.shape
    tid = None if traj_id is None else i32(traj_id)
    order = np.empty(n, dtype=np.int32)
    check(lib().bild_schedule_segments(model._h, ts._h, n, K1, iptr(seg_start), iptr(seg_state), iptr(tid),
                                       _flags(path, prefix=prefix, jump=jump), iptr(order)))
    return order


def frames_executed_fraction(model, ts, seg_start, traj_id=None, order=None, path='auto', prefix=True):
    """ share of the (task, frame) pairs of a batch that the launch runs itself (the rest comes out of the prefix table) """
    seg_start = i32(seg_start)
    n, K1 = seg_start.shape
    tid = None if traj_id is None else i32(traj_id)
    od = None if order is None else i32(order)
    tot, run = ctypes.c_double(0), ctypes.c_double(0)
    check(lib().bild_frames_executed(model._h, ts._h, n, K1, iptr(seg_start), iptr(tid), iptr(od), _flags(path, prefix=prefix),
                                     ctypes.byref(tot), ctypes.byref(run)))
    return run.value / tot.value if tot.value else 1.0


def prefix_info(ts):
    """ (bytes, build_ms) of the trajectory set's prefix table; (0, 0.0) when none has been built """
    b, ms = ctypes.c_int64(0), ctypes.c_double(0)
    check(lib().bild_prefix_info(ts._h, ctypes.byref(b), ctypes.byref(ms)))
    return b.value, ms.value


def logl_segments_device(model, ts, n, K1, d_seg_start, d_seg_state, d_traj_id, d_out, stream=0, path='auto', validate=False,
                         d_order=0, prefix=True, jump=True, split=True, states=True):
    """
    raw device pointers (ints); asynchronous on `stream` (validate=True: descriptors checked on the device first);
    d_order: device pointer of the launch order from `schedule_segments`, 0 = the order of the arrays
    """
    check(lib().bild_logl_segments_device_ordered(model._h, ts._h, n, K1, _vp(d_seg_start), _vp(d_seg_state),
                                          _vp(d_traj_id) if d_traj_id else None, _vp(d_order) if d_order else None,
                                          _flags(path, validate, prefix, jump, split, states),
                                          _vp(stream) if stream else None, _vp(d_out)))


def logl_st_device(model, ts, n, K1, d_ss, d_thetas, d_out, d_traj_id=0, stream=0, path='auto', d_status=0, split=True, states=True, tail=True):
    """
    the sampler's (s, theta) rows resident in HBM (raw device pointers: float64 n x K1, uint8 n x K1), results to d_out;
    asynchronous on `stream` (bild_logl_st_device)
    """
    check(lib().bild_logl_st_device(model._h, ts._h, n, K1, _vp(d_ss), _vp(d_thetas), _vp(d_traj_id) if d_traj_id else None,
                                    _flags(path, split=split, states=states, tail=tail), _vp(stream) if stream else None, _vp(d_out),
                                    _vp(d_status) if d_status else None))


def flop_count(model, ts, n, traj_id=None, path='auto'):
    can, exe = ctypes.c_double(0), ctypes.c_double(0)
    tid = None if traj_id is None else i32(traj_id)
    check(lib().bild_flop_count(model._h, ts._h, n, iptr(tid), PATHS[path], ctypes.byref(can), ctypes.byref(exe)))
    return can.value, exe.value


def kernel_timing(enable):
    """ False / 0: off; True / 1: events around every launch; p > 1: around every p-th launch """
    check(lib().bild_kernel_timing(int(enable)))


def kernel_timing_read():
    ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
    name = ctypes.create_string_buffer(128)
    check(lib().bild_kernel_timing_read(ctypes.byref(ms), ctypes.byref(cnt), name, 128))
    return ms.value, cnt.value, name.value.decode()


def kernel_timing_read_walk():
    """ device time and number of launches of the table-walk kernel (walk.hip) since the last call """
    ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
    check(lib().bild_kernel_timing_read_walk(ctypes.byref(ms), ctypes.byref(cnt)))
    return ms.value, cnt.value


def logl_st_status(model):
    """ waits for the model's pending to_device calls; raises if one of their rows was refused (bild_logl_st_status) """
    row = ctypes.c_int64(-1)
    check(lib().bild_logl_st_status(model._h, ctypes.byref(row)))


COMM_ID_BYTES = 128


def _choose_rccl():
    """
    The RCCL that matches the HIP runtime in the process: when the library was bound to PyTorch's runtime (see
    `_share_hip_runtime_with_torch`), PyTorch's own librccl.so beside it; else whatever the loader finds.
    """
    global _rccl_chosen
    if _rccl_chosen:
        return
    _rccl_chosen = True
    libdir = _torch_libdir
    if libdir is None and 'torch' in __import__('sys').modules:
        libdir = os.path.join(os.path.dirname(__import__('sys').modules['torch'].__file__), 'lib')
    if libdir and os.path.exists(os.path.join(libdir, 'librccl.so')) and not os.environ.get('BILD_AMD_RCCL'):
        lib().bild_comm_library(os.path.join(libdir, 'librccl.so').encode())


def comm_unique_id():
    """ 128 opaque bytes identifying a new communicator (rank 0 calls this and hands them to every rank) """
    _choose_rccl()
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    check(lib().bild_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


class CommHandle:
    """ owns a ``bild_comm*`` (an RCCL communicator on the current device) """

    def __init__(self, unique_id, world, rank):
        assert len(unique_id) == COMM_ID_BYTES
        _choose_rccl()
        self.world, self.rank = int(world), int(rank)
        self._h = _vp()
        check(lib().bild_comm_create(unique_id, self.world, self.rank, ctypes.byref(self._h)))

    def allgather(self, d_send, d_recv, n_per_rank, stream=0):
        """ raw device pointers; asynchronous on `stream` """
        check(lib().bild_comm_allgather(self._h, _vp(d_send), _vp(d_recv), int(n_per_rank), _vp(stream) if stream else None))

    def __del__(self):
        if getattr(self, '_h', None) and _lib is not None:
            _lib.bild_comm_destroy(self._h)
            self._h = None


EXCHANGE_HANDLE_BYTES = 64


class ExchangeHandle:
    """ owns a ``bild_exchange*``: the direct all-gather of a multi-GPU step (include/bild_amd.h, "the direct exchange") """

    def __init__(self, world, rank, slot_doubles):
        self.world, self.rank = int(world), int(rank)
        self._h = _vp()
        check(lib().bild_exchange_create(self.world, self.rank, int(slot_doubles), ctypes.byref(self._h)))

    def handle(self):
        buf = ctypes.create_string_buffer(EXCHANGE_HANDLE_BYTES)
        check(lib().bild_exchange_handle(self._h, buf))
        return buf.raw

    def connect(self, handles):
        """ handles: the 64-byte handles of all ranks, in rank order """
        blob = b''.join(handles)
        assert len(blob) == self.world * EXCHANGE_HANDLE_BYTES
        check(lib().bild_exchange_connect(self._h, blob))

    def allgather(self, d_send, d_recv, n_per_rank, stream=0):
        check(lib().bild_exchange_allgather(self._h, int(d_send), int(n_per_rank), int(stream) if stream else None, int(d_recv)))

    def status(self):
        """ after the stream has been waited for: raises when a peer did not deliver within the timeout """
        peer = ctypes.c_int(-1)
        check(lib().bild_exchange_status(self._h, ctypes.byref(peer)))

    def set_step(self, step):
        check(lib().bild_exchange_set_step(self._h, int(step) & 0xffffffff))

    def set_timeout(self, seconds):
        check(lib().bild_exchange_set_timeout(self._h, float(seconds)))

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                lib().bild_exchange_destroy(h)
            except Exception:  # pragma: no cover  (interpreter shutdown)
                pass


class DeviceBuffer:
    """ a plain device allocation of `n` float64 (for host programs without a GPU framework) """

    def __init__(self, n):
        self.n = int(n)
        self._p = _vp()
        check(lib().bild_device_alloc(8 * self.n, ctypes.byref(self._p)))

    @property
    def ptr(self):
        return self._p.value or 0

    def to_host(self, n=None, stream=0):
        out = np.empty(self.n if n is None else int(n), dtype=np.float64)
        check(lib().bild_device_to_host(aptr(out), self._p, 8 * out.size, _vp(stream) if stream else None))
        return out

    def __del__(self):
        if getattr(self, '_p', None) and _lib is not None:
            _lib.bild_device_free(self._p)
            self._p = None


class AmisCore:
    """
    Host-side bookkeeping of one fixed-k AMIS sampler in native code (include/bild_amd.h, "host-side AMIS
    bookkeeping"; csrc/amis_host.cpp).  No GPU involved.  `bild_amd.amis.FixedkSampler` drives it; the NumPy
    formulation of the same bookkeeping lives there as the specification.
    """
    POOL = {'logLs': 0, 'logδs': 1, 'cur_log_proposal': 2, 'log_weights': 3}

    def __init__(self, transitions, a0, logp0, concentration_brake, polarization_brake, logprior):
        trans = np.ascontiguousarray(np.asarray(transitions) != 0, dtype=np.uint8)
        self.n = trans.shape[0]
        self.k1 = len(a0)
        logp0, a0 = f64(logp0), f64(a0)
        assert logp0.shape == (self.n, self.k1)
        self._h = _vp()
        code = lib().bild_amis_create(self.k1, self.n, aptr(trans), float(concentration_brake),
                                      float(polarization_brake), float(logprior), dptr(a0), dptr(logp0),
                                      ctypes.byref(self._h))
        if code != OK:
            raise BildAmdError(code, "bild_amis_create failed")

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                lib().bild_amis_destroy(h)
            except Exception:  # pragma: no cover  (interpreter shutdown)
                pass

    def __len__(self):
        return int(lib().bild_amis_pool_size(self._h))

    def params(self, which=-1):
        a, logp = np.empty(self.k1), np.empty((self.n, self.k1))
        if lib().bild_amis_params(self._h, which, dptr(a), dptr(logp)) != OK:
            raise IndexError(which)
        return a, logp

    def pool(self, key):
        out = np.empty(len(self))
        if lib().bild_amis_pool(self._h, self.POOL[key], dptr(out)) != OK:
            raise KeyError(key)
        return out

    def restore(self, parameters, ss, thetas, arrays):
        """ load saved state into a freshly created core: further proposals [(a, logp), ...] and the pooled samples """
        Q = len(parameters)
        a = f64(np.array([p[0] for p in parameters]).reshape(Q, self.k1))
        logp = f64(np.array([p[1] for p in parameters]).reshape(Q, self.n, self.k1))
        ss = f64(np.asarray(ss).reshape(-1, self.k1))
        thetas = np.ascontiguousarray(np.asarray(thetas).reshape(-1, self.k1), dtype=np.int64)
        cols = [f64(arrays[key]) for key in ('logLs', 'logδs', 'cur_log_proposal', 'log_weights')] if len(ss) else [f64([])] * 4
        code = lib().bild_amis_restore(self._h, Q, dptr(a), dptr(logp), len(ss), dptr(ss), aptr(thetas),
                                       *(dptr(c) for c in cols))
        if code != OK:
            raise BildAmdError(code, "bild_amis_restore failed")

    def sample_traces(self, u):
        """ u: (k1, N) uniform random numbers -> (N, k1) int traces from the current proposal """
        u = f64(u)
        assert u.ndim == 2 and u.shape[0] == self.k1
        thetas = np.empty((u.shape[1], self.k1), dtype=np.int64)
        code = lib().bild_amis_sample_traces(self._h, u.shape[1], dptr(u), aptr(thetas))
        if code != OK:
            raise BildAmdError(code, "bild_amis_sample_traces failed")
        return thetas

    def use_device(self, enable=True):
        """ pooled samples in HBM, the passes of `step` on the GPU (bild_amis_use_device); for large batches per step """
        code = lib().bild_amis_use_device(self._h, 1 if enable else 0)
        if code == ERR_NO_DEVICE:
            raise NoDeviceError(code, lib().bild_amis_error(self._h).decode())
        if code != OK:
            raise BildAmdError(code, lib().bild_amis_error(self._h).decode())
        self.on_device = bool(enable)

    def step(self, ss, thetas, logLs):
        """ -> (logev, dlogev, KL); RuntimeError("Iteration did not converge") as the reference raises it """
        ss, logLs = f64(ss), f64(logLs)
        thetas = np.ascontiguousarray(thetas, dtype=np.int64)
        assert ss.shape == thetas.shape == (len(logLs), self.k1)
        ev = np.empty(3)
        code = lib().bild_amis_step(self._h, len(logLs), dptr(ss), aptr(thetas), dptr(logLs), dptr(ev))
        if code != OK:
            msg = lib().bild_amis_error(self._h).decode()
            raise RuntimeError(msg) if 'converge' in msg else BildAmdError(code, msg)
        return tuple(ev)


def _amis_step_fused(self, model, ts, ss, thetas, path='auto'):
    """
    the likelihood of the new batch and the bookkeeping of the step in one native call (bild_amis_step_fused): the samples
    go up once, nothing but partial sums comes down -> (logev, dlogev, KL)
    """
    ss = f64(ss)
    thetas = np.ascontiguousarray(thetas, dtype=np.int64)
    assert ss.shape == thetas.shape and ss.shape[1] == self.k1
    ev = np.empty(3)
    code = lib().bild_amis_step_fused(self._h, model._h, ts._h, len(ss), dptr(ss), aptr(thetas), PATHS[path], dptr(ev))
    if code != OK:
        msg = lib().bild_amis_error(self._h).decode()
        raise RuntimeError(msg) if 'converge' in msg else BildAmdError(code, msg)
    return tuple(ev)


AmisCore.step_fused = _amis_step_fused


def _amis_step_device_rng(self, model, ts, N, seed, path='auto'):
    """ a fused step whose samples are drawn on the device (bild_amis_step_device_rng) -> (logev, dlogev, KL) """
    ev = np.empty(3)
    code = lib().bild_amis_step_device_rng(self._h, model._h, ts._h, int(N), ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), PATHS[path], dptr(ev))
    if code != OK:
        msg = lib().bild_amis_error(self._h).decode()
        raise RuntimeError(msg) if 'converge' in msg else BildAmdError(code, msg)
    return tuple(ev)


def _amis_pool_samples(self):
    """ (ss, thetas) of all pooled samples, fetched from the device where fused steps left them """
    P = len(self)
    ss, thetas = np.empty((P, self.k1)), np.empty((P, self.k1), dtype=np.int64)
    if lib().bild_amis_pool_samples(self._h, dptr(ss), aptr(thetas)) != OK:
        raise BildAmdError(ERR_HIP, lib().bild_amis_error(self._h).decode())
    return ss, thetas


AmisCore.step_device_rng = _amis_step_device_rng
AmisCore.pool_samples = _amis_pool_samples


def _adopt_amis_core(handle, n, k1):
    """ an `AmisCore` around native bookkeeping that already exists (bild_run_take_core); the wrapper owns it from now on """
    core = AmisCore.__new__(AmisCore)
    core._h = handle
    core.n = n
    core.k1 = k1
    return core


class RunSettings(ctypes.Structure):
    """ bild_run_settings (include/bild_amd.h) """
    _fields_ = [('init_runs', ctypes.c_int32), ('k_lookahead', ctypes.c_int32), ('k_max', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('certainty_in_k', ctypes.c_double), ('dE', ctypes.c_double), ('N', ctypes.c_int64),
                ('concentration_brake', ctypes.c_double), ('polarization_brake', ctypes.c_double), ('max_fev', ctypes.c_int64),
                ('max_fcomplete', ctypes.c_int64), ('choice_samplesize', ctypes.c_int64)]


class RunHandle:
    """
    The inference driver (include/bild_amd.h, "the inference driver"; csrc/run_host.cpp): the adaptive-k loops of many
    trajectories, one round at a time.  `bild_amd.core.sample_many` drives it.
    """

    def __init__(self, T, transitions, settings, per_k):
        """
        T : trajectory lengths; transitions : (n, n) bool; settings : dict of the fields of `RunSettings`;
        per_k : for k = 0, 1, ...: (logp0 (n, k+1), logprior, n_total, traces (m, k+1) int or None)
        """
        trans = np.ascontiguousarray(np.asarray(transitions) != 0, dtype=np.uint8)
        self.n = trans.shape[0]
        self.n_traj = len(T)
        Ts = i32(np.asarray(T))
        st = RunSettings(**settings)
        logp0 = f64(np.concatenate([np.asarray(p[0], dtype=np.float64).reshape(-1) for p in per_k]))
        logprior = f64([p[1] for p in per_k])
        n_total = f64([p[2] for p in per_k])
        n_traces = np.array([0 if p[3] is None else len(p[3]) for p in per_k], dtype=np.int64)
        given = [np.asarray(p[3], dtype=np.int32).reshape(-1) for p in per_k if p[3] is not None and len(p[3])]
        traces = np.ascontiguousarray(np.concatenate(given) if given else np.zeros(1, dtype=np.int32), dtype=np.int32)
        self._h = _vp()
        check(lib().bild_run_create(self.n_traj, iptr(Ts), self.n, aptr(trans), ctypes.addressof(st), len(per_k), dptr(logp0),
                                    dptr(logprior), dptr(n_total), aptr(n_traces), iptr(traces), ctypes.byref(self._h)))
        self._counts = np.zeros(8, dtype=np.int64)

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                lib().bild_run_destroy(h)
            except Exception:  # pragma: no cover  (interpreter shutdown)
                pass

    def _check(self, code):
        if code != OK:
            msg = lib().bild_run_error(self._h).decode() or lib().bild_last_error().decode()
            raise (NoDeviceError if code == ERR_NO_DEVICE else BildAmdError)(code, msg)

    def plan(self):
        """
        plan the next round -> (counts, shapes): counts = (gammas, uniforms, normals, rows, trajectories running, AMIS steps,
        trajectories failed so far),
        shapes = the gamma shape parameters (a view of the driver's buffer, valid until the next call)
        """
        ptr = _vp()
        self._check(lib().bild_run_plan(self._h, aptr(self._counts), ctypes.byref(ptr)))
        n = int(self._counts[0])
        shapes = np.frombuffer((ctypes.c_double * n).from_address(ptr.value), dtype=np.float64) if n else np.empty(0)
        return self._counts, shapes

    def round(self, model, ts, gammas, uniforms, normals, path='auto'):
        """ the planned round on the GPU: rows, ONE likelihood call over all of them, bookkeeping (bild_run_round) """
        self._check(lib().bild_run_round(self._h, model._h, ts._h, PATHS[path], dptr(gammas), dptr(uniforms), dptr(normals)))

    def stage(self, gammas, uniforms):
        """ -> (ss (n, K1), thetas (n, K1) int64, traj_id (n,)): the round's candidate rows, for a caller that evaluates them itself """
        self._check(lib().bild_run_stage(self._h, dptr(gammas), dptr(uniforms)))
        n, K1 = ctypes.c_int64(0), ctypes.c_int(0)
        p_ss, p_th, p_id = _vp(), _vp(), _vp()
        self._check(lib().bild_run_rows(self._h, ctypes.byref(n), ctypes.byref(K1), ctypes.byref(p_ss), ctypes.byref(p_th), ctypes.byref(p_id)))
        n, K1 = n.value, K1.value
        if n == 0:
            return np.empty((0, K1)), np.empty((0, K1), dtype=np.int64), np.empty(0, dtype=np.int32)
        ss = np.frombuffer((ctypes.c_double * (n * K1)).from_address(p_ss.value), dtype=np.float64).reshape(n, K1)
        thetas = np.frombuffer((ctypes.c_int64 * (n * K1)).from_address(p_th.value), dtype=np.int64).reshape(n, K1)
        traj_id = np.frombuffer((ctypes.c_int32 * n).from_address(p_id.value), dtype=np.int32)
        return ss, thetas, traj_id

    def finish(self, logLs, normals):
        logLs = f64(logLs)
        self._check(lib().bild_run_finish(self._h, dptr(logLs), dptr(normals)))

    # -- results ---------------------------------------------------------------------------------------------
    def traj_info(self, j):
        """ -> (state, kind of failure, samplers, log rows, width, message) """
        info = np.zeros(5, dtype=np.int64)
        msg = ctypes.c_char_p()
        self._check(lib().bild_run_traj_info(self._h, j, aptr(info), ctypes.byref(msg)))
        return int(info[0]), int(info[1]), int(info[2]), int(info[3]), int(info[4]), (msg.value or b'').decode()

    def traj_log(self, j, rows, width):
        k, flags = np.zeros(rows, dtype=np.int32), np.zeros(rows, dtype=np.int32)
        i_la, pk, kld = np.zeros(rows), np.zeros((rows, width)), np.zeros((rows, width))
        self._check(lib().bild_run_traj_log(self._h, j, iptr(k), iptr(flags), dptr(i_la), dptr(pk), dptr(kld)))
        return k, flags, i_la, pk, kld

    def sampler_info(self, j, k):
        """ -> (kind, exhausted, steps, evidences, enumerated rows) """
        info = np.zeros(5, dtype=np.int64)
        self._check(lib().bild_run_sampler_info(self._h, j, k, aptr(info)))
        return int(info[0]), bool(info[1]), int(info[2]), int(info[3]), int(info[4])

    def sampler_data(self, j, k, n_ev, rows):
        ev = np.zeros((n_ev, 3))
        ss, th, logL = np.zeros((rows, k + 1)), np.zeros((rows, k + 1), dtype=np.int64), np.zeros(rows)
        self._check(lib().bild_run_sampler_data(self._h, j, k, dptr(ev), dptr(ss), aptr(th), dptr(logL)))
        return ev, ss, th, logL

    def take_core(self, j, k):
        h = _vp()
        self._check(lib().bild_run_take_core(self._h, j, k, ctypes.byref(h)))
        return _adopt_amis_core(h, self.n, k + 1) if h.value else None

    def totals(self):
        """ -> (rounds, likelihood evaluations, host threads) """
        t = np.zeros(3, dtype=np.int64)
        self._check(lib().bild_run_totals(self._h, aptr(t)))
        return int(t[0]), int(t[1]), int(t[2])


def choice_counts(rvs, mu, dmu, dE, omit=None, want_dn=True):
    """
    histograms of the "best k" over the rows of the common random sample (bild_choice_counts):
    (n0, Dn or None, n_omit or None)
    """
    rvs, mu, dmu = f64(rvs), f64(mu), f64(dmu)
    samplesize, kmax = rvs.shape
    n0 = np.zeros(kmax, dtype=np.int64)
    dn = np.zeros((kmax, kmax), dtype=np.int64) if want_dn else None
    flags = n_omit = None
    if omit is not None:
        flags = np.zeros(kmax, dtype=np.uint8)
        flags[omit] = 1
        n_omit = np.zeros(kmax, dtype=np.int64)
    vp = lambda a: None if a is None else aptr(a)
    code = lib().bild_choice_counts(samplesize, kmax, dptr(rvs), dptr(mu), dptr(dmu), float(dE), vp(flags), vp(n0), vp(dn), vp(n_omit))
    if code != OK:
        raise BildAmdError(code, "bild_choice_counts failed")
    return n0, dn, n_omit


def interval_marginals(seg_start, seg_state, w, n, T):
    """ (n, T) weighted state occupancy per frame of the profiles (seg_start, seg_state) with weights w """
    seg_start, seg_state, w = i32(seg_start), i32(seg_state), f64(w)
    P, k1 = seg_state.shape
    post = np.empty((n, T), dtype=np.float64)
    code = lib().bild_interval_marginals(P, k1, n, T, iptr(seg_start), iptr(seg_state), dptr(w), dptr(post))
    if code != OK:
        raise BildAmdError(code, "bild_interval_marginals failed")
    return post
