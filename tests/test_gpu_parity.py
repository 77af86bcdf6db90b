"""
GPU parity tests (run with -m gpu on an MI355X): the HIP kernels, called through the C ABI,
against (a) the committed golden vectors of the reference's own kernels and (b) the CPU
oracle on seeded inputs.

Tolerance: |delta logL| < 1e-8 absolute (BASELINE.json north_star), fp64.
"""
import os

import numpy as np
import pytest

import goldens
import helpers as H

pytestmark = pytest.mark.gpu

TOL = 1e-8
PATHS = ['modal', 'dense']


def _model_from_golden(g, path, reduce=True):
    import bild_amd
    m = bild_amd.MultiStateRouse.from_arrays(g['B'], g['G'], g['Sig'], g['M0'], g['C0'], g['w'],
                                             localization_error=g['localization_error'], path=path)
    if not reduce:
        from bild_amd import _lib
        a = m.arrays()
        m._handle = _lib.ModelHandle(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], m.measurement, reduce=False)
    return m


def test_native_library_is_loaded(built_lib):
    from bild_amd import _lib
    assert _lib.device_count() >= 1
    with open('/proc/self/maps') as f:
        assert 'libbild_amd.so' in f.read()


@pytest.mark.parametrize('name', goldens.names())
@pytest.mark.parametrize('path', PATHS)
@pytest.mark.parametrize('reduce', [True, False])
def test_goldens(built_lib, name, path, reduce):
    g = goldens.load(name)
    m = _model_from_golden(g, path, reduce)
    got = m.logL_batch(g['states'], g['x'])
    assert got.shape == (len(g['states']),)
    for key in ('logL_ref_numpy', 'logL_ref_cython'):
        ref = g[key]
        ok = ~np.isnan(ref)
        if np.any(ok):
            assert np.max(np.abs(got[ok] - ref[ok])) < TOL, (name, path, key, got, ref)
    # single-profile entry point (MultiStateModel.logL contract: returns a python float)
    one = m.logL(H.ProfileView(g['states'][0]), g['x'])
    assert isinstance(one, float) and abs(one - got[0]) < 1e-12


def test_reference_unittest_behaviours(built_lib):
    """ behavioural pins of reference tests/test_bild.py:135-151 on its own fixture """
    import bild_amd
    traj = bild_amd.Trajectory([1, 2, np.nan, 4], localization_error=[0.5])
    profile = bild_amd.Loopingprofile([1, 1, 0, 0])
    model = bild_amd.MultiStateRouse(20, 1, 5, d=1)
    logL = model.logL(profile, traj)
    assert -100 < logL < 0                                      # :138
    traj_noerr = bild_amd.Trajectory([1, 2, np.nan, 4])
    with pytest.raises(ValueError):                             # :140-143
        model.logL(profile, traj_noerr)
    model2 = bild_amd.MultiStateRouse(20, 1, 5, d=1, localization_error=0.5)
    assert model2.logL(profile, traj_noerr) == logL             # :145-148 (identical, not just close)


@pytest.mark.parametrize('path', PATHS)
@pytest.mark.parametrize('S,T,k,miss', [(2, 200, 2, 'none'), (2, 333, 5, 'iid'), (3, 400, 4, 'bursty'), (2, 97, 0, 'none')])
def test_sampler_batch_vs_oracle(built_lib, path, S, T, k, miss):
    """ FixedkSampler.logL(ss, thetas) == oracle on the st2profile-expanded profiles """
    import bild_amd
    from oracle import oracle
    rng = np.random.default_rng(100 * S + T + k)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, looppositions=H.LOOPS[S], localization_error=0.1, path=path)
    truth = H.random_profile(rng, T, S, max(T // 5, 2))
    traj = model.trajectory_from_loopingprofile(truth, missing_frames=H.missing_mask(rng, T, miss), rng=rng)
    n = 300
    ss, thetas = H.candidate_profiles(rng, n, k, S)
    sampler = bild_amd.FixedkSampler(traj, model, k=k, N=n)
    got = sampler.logL(ss, thetas)
    states = H.expand(ss, thetas, T)
    # (the encoding itself is pinned against the REFERENCE's st2profile output in test_st_seam_against_reference_vectors
    # below and, without a GPU, in tests/test_profiles.py; this line only ties the sampler method to the batch encoding)
    assert np.array_equal(sampler.st2profile(ss[7], thetas[7])[:], states[7])
    want = oracle.logl_batch(model.arrays(), model.measurement, model.localization_error, traj[:], states)
    assert got.dtype == np.float64 and got.shape == (n,)
    assert np.max(np.abs(got - want)) < TOL


@pytest.mark.parametrize('N', [2, 4, 7, 8, 11, 16, 20, 24, 27, 31, 32, 48, 64])
def test_chain_lengths(built_lib, N):
    import bild_amd
    from oracle import oracle
    rng = np.random.default_rng(N)
    T = 120
    for reduce in (True, False):
        if not reduce and N > 32:
            continue            # beyond the compiled envelope without the reduction (checked in test_error_paths)
        model = bild_amd.MultiStateRouse(N, 1, 2, d=2, localization_error=[0.1, 0.2])
        if not reduce:
            from bild_amd import _lib
            a = model.arrays()
            model._handle = _lib.ModelHandle(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], model.measurement, reduce=False)
        truth = H.random_profile(rng, T, 2, 30)
        traj = model.trajectory_from_loopingprofile(truth, missing_frames=0.1, rng=rng)
        ss, thetas = H.candidate_profiles(rng, 40, 3, 2)
        states = H.expand(ss, thetas, T)
        want = oracle.logl_batch(model.arrays(), model.measurement, model.localization_error, traj[:], states)
        for path in PATHS:
            model.path = path
            got = model.logL_st_batch(ss, thetas, traj)
            assert np.max(np.abs(got - want)) < TOL, (N, reduce, path)


@pytest.mark.parametrize('case', ['N48_full', 'N100_end2end', 'N90_3state', 'N128_full', 'N40_force', 'N256_end2end'])
def test_long_chains(built_lib, case):
    """
    more than 32 effective modes: the LDS-resident kernel (wide.hip) against the oracle -- missing frames,
    two localization errors, three states, an external force, the largest supported size, several trajectories
    """
    import bild_amd
    from bild_amd import _lib
    from oracle import oracle
    rng = np.random.default_rng(len(case) + 7)
    S, T, nprof, k, reduce, err, d = 2, 90, 12, 3, True, [0.1, 0.2], 2
    if case == 'N48_full':
        N, reduce = 48, False
    elif case == 'N100_end2end':
        N = 100
    elif case == 'N90_3state':
        N, S, d, err = 90, 3, 3, [0.1, 0.1, 0.3]
    elif case == 'N128_full':
        N, reduce, T, nprof, d, err = 128, False, 40, 4, 1, [0.15]
    elif case == 'N40_force':
        N, reduce, d, err = 40, False, 3, [0.1, 0.1, 0.2]
    else:
        N, T, nprof, d, err = 256, 40, 4, 1, [0.15]
    model = bild_amd.MultiStateRouse(N, 1, 2, d=d, looppositions=H.LOOPS[S] if S == 3 else (None, (0, -1)),
                                     localization_error=err)
    if case == 'N40_force':
        for mi, mod in enumerate(model.models):
            mod.F[0, :] = [0.5, -0.25, 0.1 * (mi + 1)]
            mod.F[-1, :] = [-0.5, 0.25, -0.1 * (mi + 1)]
            mod.F[7, 1] = 0.3
            mod.update_dynamics()
    a = model.arrays()
    model._handle = _lib.ModelHandle(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], model.measurement, reduce=reduce)
    assert model.handle().query(_lib.Q_NEFF) > 32
    trajs = [model.trajectory_from_loopingprofile(H.random_profile(rng, T + 7 * j, S, 25), missing_frames=0.1 * j, rng=rng)
             for j in range(2)]
    for j, traj in enumerate(trajs):
        ss, thetas = H.candidate_profiles(rng, nprof, k, S)
        want = oracle.logl_batch(a, model.measurement, model.localization_error, traj[:], H.expand(ss, thetas, len(traj)))
        got = model.logL_st_batch(ss, thetas, traj)
        assert np.max(np.abs(got - want)) < TOL, (case, j, np.max(np.abs(got - want)))


def test_multi_trajectory_batch(built_lib):
    """ samples spread over several trajectories of different length / noise / masks """
    import bild_amd
    from oracle import oracle
    from bild_amd.profiles import segments_from_st
    rng = np.random.default_rng(5)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3)
    trajs, seg_start, seg_state, tid, want = [], [], [], [], []
    K = 4
    for j, (T, err) in enumerate([(150, 0.1), (90, [0.1, 0.1, 0.3]), (211, 0.05), (64, [0.2, 0.1, 0.05])]):
        truth = H.random_profile(rng, T, 2, 40)
        tr = model.trajectory_from_loopingprofile(truth, localization_error=err, missing_frames=0.05 * j, rng=rng)
        trajs.append(tr)
        ss, thetas = H.candidate_profiles(rng, 25, K, 2)
        a, b = segments_from_st(ss, thetas, T)
        seg_start.append(a)
        seg_state.append(b)
        tid += [j] * 25
        want.append(oracle.logl_batch(model.arrays(), model.measurement, tr.localization_error, tr[:], H.expand(ss, thetas, T)))
    perm = rng.permutation(100)
    seg_start = np.concatenate(seg_start)[perm]
    seg_state = np.concatenate(seg_state)[perm]
    tid = np.asarray(tid)[perm]
    want = np.concatenate(want)[perm]
    for path in PATHS:
        model.path = path
        got = model.logL_segments(seg_start, seg_state, trajs, tid)
        assert np.max(np.abs(got - want)) < TOL


def test_full_size_properties(built_lib):
    """
    BASELINE configs[1] size (10k samples x T=1000): properties that do not need the oracle
    at full size -- both kernel paths agree, permutation equivariance, and a 64-sample
    spot check against the oracle.
    """
    import bild_amd
    from oracle import oracle
    rng = np.random.default_rng(42)
    T, n, k = 1000, 10000, 4
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    truth = H.random_profile(rng, T, 2, 200)
    traj = model.trajectory_from_loopingprofile(truth, rng=rng)
    ss, thetas = H.candidate_profiles(rng, n, k, 2)
    model.path = 'modal'
    a = model.logL_st_batch(ss, thetas, traj)
    model.path = 'dense'
    b = model.logL_st_batch(ss, thetas, traj)
    assert np.all(np.isfinite(a)) and np.max(np.abs(a - b)) < TOL
    perm = rng.permutation(n)
    model.path = 'modal'
    c = model.logL_st_batch(ss[perm], thetas[perm], traj)
    assert np.array_equal(c, a[perm])          # bit-identical regardless of placement in the batch
    pick = rng.choice(n, 64, replace=False)
    want = oracle.logl_batch(model.arrays(), model.measurement, model.localization_error, traj[:],
                             H.expand(ss[pick], thetas[pick], T))
    assert np.max(np.abs(a[pick] - want)) < TOL


def test_error_paths(built_lib):
    import bild_amd
    from bild_amd import _lib
    model = bild_amd.MultiStateRouse(8, 1, 2, d=2, localization_error=0.1)
    traj = bild_amd.Trajectory(np.zeros((10, 2)))
    with pytest.raises(_lib.BildAmdError):     # state index out of range
        model.logL_batch(np.full((1, 10), 5), traj)
    with pytest.raises(AssertionError):        # wrong spatial dimension (reference asserts shapes, pyx:165-166)
        model.logL_batch(np.zeros((1, 10), int), bild_amd.Trajectory(np.zeros((10, 3))))
    assert model.logL_batch(np.zeros((0, 10), int), traj).shape == (0,)
    big = bild_amd.MultiStateRouse(130, 1, 2, d=1, localization_error=0.1)    # 130 modes after NO reduction
    a = big.arrays()
    with pytest.raises(_lib.BildAmdError):                                      # outside the envelope: loud
        _lib.ModelHandle(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], big.measurement, reduce=False)
    long = bild_amd.MultiStateRouse(80, 1, 2, d=1, localization_error=0.1, path='dense')   # 40 modes: modal only
    with pytest.raises(_lib.BildAmdError):
        long.logL_batch(np.zeros((1, 10), int), bild_amd.Trajectory(np.zeros((10, 1))))


def test_device_pointer_entry_point(built_lib):
    """ bild_logl_segments_device: torch tensors in HBM, launch on torch's current stream (what bench.py times) """
    import torch
    import bild_amd
    from bild_amd import _lib
    from bild_amd.profiles import segments_from_st
    rng = np.random.default_rng(12)
    T, n, k = 150, 777, 3
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=[0.1, 0.1, 0.2])   # d* = 2: exercises the reduce kernel
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 40), missing_frames=0.1, rng=rng)
    ss, thetas = H.candidate_profiles(rng, n, k, 2)
    want = model.logL_st_batch(ss, thetas, traj)                                # host-buffer entry point
    a, b = segments_from_st(ss, thetas, T)
    dev = torch.device('cuda', 0)
    d_a, d_b = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    out = torch.full((n,), float('nan'), dtype=torch.float64, device=dev)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        _lib.logl_segments_device(model.handle(), model.trajset(traj), n, k + 1, d_a.data_ptr(), d_b.data_ptr(), 0,
                                  out.data_ptr(), stream=stream.cuda_stream)
    stream.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)                              # same kernel, same bits


def test_large_batch_and_threads(built_lib):
    """ a batch that wraps the grid-stride loop many times, and concurrent host-side callers on one model """
    import threading
    import bild_amd
    from oracle import oracle
    rng = np.random.default_rng(21)
    T, k = 64, 2
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 20), rng=rng)
    n = 600_000
    ss, thetas = H.candidate_profiles(rng, n, k, 2)
    got = model.logL_st_batch(ss, thetas, traj)
    pick = rng.choice(n, 200, replace=False)
    want = oracle.logl_batch(model.arrays(), model.measurement, model.localization_error, traj[:], H.expand(ss[pick], thetas[pick], T))
    assert np.all(np.isfinite(got)) and np.max(np.abs(got[pick] - want)) < TOL
    # identical profiles give identical results wherever they sit in the batch
    assert np.array_equal(model.logL_st_batch(ss[pick], thetas[pick], traj), got[pick])

    results = {}

    def worker(i):
        sl = slice(i * 1000, (i + 1) * 1000)
        results[i] = model.logL_st_batch(ss[sl], thetas[sl], traj)
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for i in range(6):
        assert np.array_equal(results[i], got[i * 1000:(i + 1) * 1000])


class _OracleRouse:
    """ CPU stand-in with the reference kernel semantics (oracle), for end-to-end comparisons """

    def __init__(self, model):
        self.m = model
        self.transitions, self.nStates, self.d = model.transitions, model.nStates, model.d

    def logL(self, profile, traj):
        from oracle import oracle
        return oracle.logl(self.m.arrays(), self.m.measurement, self.m._get_noise(traj), traj[:], np.asarray(profile[:]))


def test_amis_steps_gpu_equal_cpu_oracle(built_lib):
    """ whole AMIS iterations: GPU likelihood vs oracle likelihood, same random stream """
    import bild_amd
    rng = np.random.default_rng(3)
    T = 60
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    truth = H.random_profile(rng, T, 2, 20)
    traj = model.trajectory_from_loopingprofile(truth, missing_frames=0.05, rng=rng)
    runs = []
    for mdl in (model, _OracleRouse(model)):
        np.random.seed(99)
        sampler = bild_amd.FixedkSampler(traj, mdl, k=2, N=50, max_fcomplete=10)
        for _ in range(3):
            assert sampler.step()
        runs.append(sampler)
    a, b = runs
    for sa, sb in zip(a.samples, b.samples):
        assert np.array_equal(sa['thetas'], sb['thetas'])
        assert np.max(np.abs(sa['logLs'] - sb['logLs'])) < TOL
    assert np.allclose(np.array(a.evidences), np.array(b.evidences), rtol=0, atol=1e-7)
    assert np.array_equal(a.MAP_profile()[:], b.MAP_profile()[:])


def test_sample_many_and_postproc_gpu(built_lib):
    """ config-5 style: the adaptive-k loop for several trajectories with fused launches, then boundary polishing """
    import bild_amd
    from bild_amd import postproc
    rng = np.random.default_rng(8)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    trajs, truths = [], []
    for T in (40, 55, 70):
        truth = H.random_profile(rng, T, 2, 18)
        truths.append(truth)
        trajs.append(model.trajectory_from_loopingprofile(truth, rng=rng))
    kw = dict(init_runs=3, k_max=4, sampler_kw={'N': 40, 'max_fev': 800, 'max_fcomplete': 50}, choice_kw={'samplesize': 1000})
    np.random.seed(5)
    fused = bild_amd.sample_many(trajs, model, **kw)
    np.random.seed(5)
    again = bild_amd.sample_many(trajs, model, **kw)
    oracle_model = _OracleRouse(model)
    for res, res2, traj in zip(fused, again, trajs):
        assert np.array_equal(res.evidence, res2.evidence)           # deterministic
        assert np.all(np.isfinite(res.evidence[:3]))
        prof = res.best_profile()
        # GPU single-profile likelihood == oracle on the inferred profile
        assert abs(model.logL(prof, traj) - oracle_model.logL(prof, traj)) < TOL
        # boundary polishing (2k+1 profiles per iteration as one GPU batch) agrees with the CPU oracle
        try:
            pg = postproc.optimize_boundary(prof, traj, model)
            pc = postproc.optimize_boundary(prof, traj, oracle_model)
            assert pg == pc
            assert model.logL(pg, traj) >= model.logL(prof, traj) - 1e-9
        except postproc.BoundaryEliminationError:
            with pytest.raises(postproc.BoundaryEliminationError):
                postproc.optimize_boundary(prof, traj, oracle_model)


def _spot_check(model, trajs, seg_start, seg_state, tid, got, rng, n_check, Ts):
    from oracle import oracle
    from bild_amd.profiles import states_from_segments
    pick = rng.choice(len(got), n_check, replace=False)
    worst = 0.0
    for i in pick:
        j = int(tid[i])
        st = states_from_segments(seg_start[i:i + 1], seg_state[i:i + 1], Ts[j])
        want = oracle.logl(model.arrays(), model.measurement, model._get_noise(trajs[j]), trajs[j][:], st[0])
        worst = max(worst, abs(got[i] - want))
    return worst


def test_baseline_config3_shape(built_lib):
    """
    BASELINE configs[2] per-GPU share: 32 trajectories x 1000 samples, T = 1000, 2-state, one launch over the
    device-resident trajectory set (the full config is 256 trajectories over 8 GPUs, sharded by trajectory).
    """
    import bild_amd
    from bild_amd.profiles import segments_from_st
    from bild_amd.dist import shard_by_trajectory
    rng = np.random.default_rng(33)
    T, n_traj, per, k = 1000, 32, 1000, 4
    owners = shard_by_trajectory(np.full(256, T), np.full(256, per), 8)
    assert all(len(o) == n_traj for o in owners)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    trajs = [model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 200), rng=rng) for _ in range(n_traj)]
    ss, thetas = H.candidate_profiles(rng, n_traj * per, k, 2)
    seg_start, seg_state = segments_from_st(ss, thetas, T)
    tid = np.repeat(np.arange(n_traj), per).astype(np.int32)
    got = model.logL_segments(seg_start, seg_state, trajs, tid)
    assert got.shape == (n_traj * per,) and np.all(np.isfinite(got))
    assert _spot_check(model, trajs, seg_start, seg_state, tid, got, rng, 48, [T] * n_traj) < TOL
    # the same samples, shuffled across trajectories: placement in the batch must not matter
    perm = rng.permutation(len(got))
    again = model.logL_segments(seg_start[perm], seg_state[perm], trajs, tid[perm])
    assert np.array_equal(again, got[perm])


def test_baseline_config4_shape(built_lib):
    """
    BASELINE configs[3]: 3-state model, T = 2000, mixed missing-frame masks (none / iid 10 % / bursty 30 %,
    frame 0 missing in half of them), several trajectories in one batch.
    """
    import bild_amd
    from bild_amd.profiles import segments_from_st
    rng = np.random.default_rng(44)
    T, per, k = 2000, 1250, 5
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, looppositions=H.LOOPS[3], localization_error=0.1)
    trajs = []
    for j, kind in enumerate(['none', 'iid', 'bursty', 'bursty']):
        miss = H.missing_mask(rng, T, kind)
        if j % 2 == 1:
            miss = np.union1d(miss, [0])
        trajs.append(model.trajectory_from_loopingprofile(H.random_profile(rng, T, 3, 300), missing_frames=miss, rng=rng))
    ss, thetas = H.candidate_profiles(rng, len(trajs) * per, k, 3)
    seg_start, seg_state = segments_from_st(ss, thetas, T)
    tid = np.repeat(np.arange(len(trajs)), per).astype(np.int32)
    for path in PATHS:
        model.path = path
        got = model.logL_segments(seg_start, seg_state, trajs, tid)
        assert np.all(np.isfinite(got))
        assert _spot_check(model, trajs, seg_start, seg_state, tid, got, rng, 24, [T] * len(trajs)) < TOL


@pytest.mark.parametrize('case', ['long_T', 'many_switches', 'many_tasks'])
def test_extreme_shapes(built_lib, case):
    """ maximum sizes: a very long trajectory, more switches than a register-resident table would hold, a huge batch """
    import bild_amd
    from oracle import oracle
    rng = np.random.default_rng({'long_T': 1, 'many_switches': 2, 'many_tasks': 3}[case])
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    if case == 'long_T':
        T, n, k, check = 60_000, 48, 6, 6
    elif case == 'many_switches':
        T, n, k, check = 2_000, 256, 400, 24
    else:
        T, n, k, check = 12, 3_000_000, 2, 200
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, max(T // 7, 2)), missing_frames=0.05, rng=rng)
    ss, thetas = H.candidate_profiles(rng, n, k, 2)
    got = model.logL_st_batch(ss, thetas, traj)
    assert got.shape == (n,) and np.all(np.isfinite(got))
    pick = rng.choice(n, check, replace=False)
    want = oracle.logl_batch(model.arrays(), model.measurement, model.localization_error, traj[:],
                             H.expand(ss[pick], thetas[pick], T))
    # north_star quotes |delta| < 1e-8 at T = 1000; the running sums of the T = 60 000 case are 60x longer and the
    # bar is scaled with T for it (6e-7).  Observed values are printed (run with -s) and recorded in tests/README.md.
    worst = np.max(np.abs(got[pick] - want))
    from bild_amd import _lib
    exact = _lib.logl_st(model.handle(), model.trajset(traj), ss[pick], thetas[pick], jump=False)
    print(f"extreme shape {case}: max|delta| vs oracle = {worst:.2e} (frame by frame: {np.max(np.abs(exact - want)):.2e}) "
          f"on |logL| ~ {np.max(np.abs(want)):.1e}")
    assert worst < TOL * max(1, T // 1000), case
    # what the tables add on top of the frame-by-frame run is bounded on its own (observed: equal to three digits in all
    # three cases, tests/README.md), and the deviation from the oracle is a RELATIVE one: 1.3e-12 of |logL| at T = 60 000
    assert np.max(np.abs(got[pick] - exact)) < 1e-9 * max(1, T // 1000), case
    assert worst < 3e-12 * max(np.max(np.abs(want)), 1e4), case


def test_model_survives_pickling_and_copying(built_lib):
    """ device handles are dropped from the pickled state and recreated on first use """
    import copy
    import pickle
    import bild_amd
    rng = np.random.default_rng(8)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, 150, 2, 40), rng=rng)
    ss, thetas = H.candidate_profiles(rng, 33, 3, 2)
    want = model.logL_st_batch(ss, thetas, traj)
    for clone in (pickle.loads(pickle.dumps(model)), copy.deepcopy(model)):
        assert clone._handle is None
        assert np.array_equal(clone.logL_st_batch(ss, thetas, traj), want)
    assert np.array_equal(model.logL_st_batch(ss, thetas, traj), want)      # the original is untouched


@pytest.mark.parametrize('case', ['N8', 'N24', 'N20_full', 'N16_3state_full', 'N12_force_full'])
def test_dense_path_on_the_matrix_pipe(built_lib, case):
    """
    dense_mfma.hip (chains that tile into 4x4 blocks): four tasks per wavefront with different trajectories, lengths,
    missing frames, states and switch times; against the oracle and against the vector formulation of the same path
    """
    import bild_amd
    from bild_amd import _lib
    from bild_amd.profiles import segments_from_st
    from oracle import oracle
    rng = np.random.default_rng(len(case))
    N, S, reduce, d, err = {'N8': (8, 2, True, 3, 0.1), 'N24': (24, 2, True, 2, [0.1, 0.3]), 'N20_full': (20, 2, False, 3, 0.1),
                            'N16_3state_full': (16, 3, False, 3, [0.1, 0.1, 0.2]), 'N12_force_full': (12, 2, False, 3, 0.2)}[case]
    model = bild_amd.MultiStateRouse(N, 1, 3, d=d, looppositions=H.LOOPS[S] if S == 3 else (None, (0, -1)),
                                     localization_error=err, path='dense')
    if case == 'N12_force_full':
        for mi, mod in enumerate(model.models):
            mod.F[0, :] = [0.5, -0.25, 0.1 * (mi + 1)]
            mod.F[-1, :] = [-0.5, 0.25, -0.1 * (mi + 1)]
            mod.update_dynamics()
    a = model.arrays()
    model._handle = _lib.ModelHandle(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], model.measurement, reduce=reduce)
    assert model.handle().query(_lib.Q_NP) % 4 == 0
    trajs, seg_start, seg_state, tid, want = [], [], [], [], []
    K = 5
    for j, T in enumerate([150, 37, 211, 64, 5]):
        tr = model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, max(T // 4, 2)), missing_frames=0.07 * j if T > 8 else None, rng=rng)
        trajs.append(tr)
        ss, thetas = H.candidate_profiles(rng, 21, K, S)
        sa, sb = segments_from_st(ss, thetas, T)
        seg_start.append(sa)
        seg_state.append(sb)
        tid += [j] * 21
        want.append(oracle.logl_batch(a, model.measurement, model.localization_error, tr[:], H.expand(ss, thetas, T)))
    perm = rng.permutation(len(tid))                       # neighbouring tasks of a wave differ in everything
    seg_start, seg_state = np.concatenate(seg_start)[perm], np.concatenate(seg_state)[perm]
    tid, want = np.asarray(tid)[perm], np.concatenate(want)[perm]
    _lib.kernel_timing(True)
    got = model.logL_segments(seg_start, seg_state, trajs, tid)
    _lib.kernel_timing(False)
    assert _lib.kernel_timing_read()[2] == 'logl_dense_mfma_kernel'
    assert np.max(np.abs(got - want)) < TOL
    os.environ['BILD_DENSE_VALU'] = '1'
    try:
        valu = model.logL_segments(seg_start, seg_state, trajs, tid)
    finally:
        del os.environ['BILD_DENSE_VALU']
    assert np.max(np.abs(got - valu)) < 1e-9


@pytest.mark.parametrize('case', ['N72', 'N36_full_3state', 'N80_force'])
def test_modal_recursion_on_tile_registers(built_lib, case):
    """
    modal_mfma.hip (33-40 modes): four tasks per wavefront with different trajectories, lengths, missing frames and
    switch times (a switch of one task makes the whole wave run the basis change, the others with the identity)
    """
    import bild_amd
    from bild_amd import _lib
    from bild_amd.profiles import segments_from_st
    from oracle import oracle
    rng = np.random.default_rng(len(case) + 3)
    N, S, reduce, d, err = {'N72': (72, 2, True, 3, 0.1), 'N36_full_3state': (36, 3, False, 2, [0.1, 0.25]),
                            'N80_force': (80, 2, True, 3, [0.1, 0.1, 0.2])}[case]
    model = bild_amd.MultiStateRouse(N, 1, 3, d=d, looppositions=H.LOOPS[S] if S == 3 else (None, (0, -1)), localization_error=err)
    if case == 'N80_force':
        for mi, mod in enumerate(model.models):
            mod.F[0, :] = [0.5, -0.25, 0.1 * (mi + 1)]
            mod.F[-1, :] = [-0.5, 0.25, -0.1 * (mi + 1)]
            mod.update_dynamics()
    a = model.arrays()
    model._handle = _lib.ModelHandle(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], model.measurement, reduce=reduce)
    assert 32 < model.handle().query(_lib.Q_NEFF) <= 40
    trajs, seg_start, seg_state, tid, want = [], [], [], [], []
    K = 5
    for j, T in enumerate([90, 37, 121, 64, 5]):
        tr = model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, max(T // 4, 2)), missing_frames=0.07 * j if T > 8 else None, rng=rng)
        trajs.append(tr)
        ss, thetas = H.candidate_profiles(rng, 9, K, S)
        sa, sb = segments_from_st(ss, thetas, T)
        seg_start.append(sa)
        seg_state.append(sb)
        tid += [j] * 9
        want.append(oracle.logl_batch(a, model.measurement, model.localization_error, tr[:], H.expand(ss, thetas, T)))
    perm = rng.permutation(len(tid))
    seg_start, seg_state = np.concatenate(seg_start)[perm], np.concatenate(seg_state)[perm]
    tid, want = np.asarray(tid)[perm], np.concatenate(want)[perm]
    _lib.kernel_timing(True)
    got = model.logL_segments(seg_start, seg_state, trajs, tid)
    _lib.kernel_timing(False)
    assert _lib.kernel_timing_read()[2] == 'logl_modal_mfma_kernel'
    assert np.max(np.abs(got - want)) < TOL


def test_torch_after_the_library(built_lib):
    """
    A PyTorch-ROCm wheel brings its own HIP runtime; the library binds to it when torch is installed, so that
    ``torch.cuda`` still works when it is first used AFTER the library (it found "No HIP GPUs" otherwise).
    """
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n"
              "import numpy as np, helpers as H, bild_amd\n"
              "assert 'torch' not in sys.modules\n"
              "m = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)\n"
              "rng = np.random.default_rng(0)\n"
              "tr = m.trajectory_from_loopingprofile(H.random_profile(rng, 50, 2, 20), rng=rng)\n"
              "ss, th = H.candidate_profiles(rng, 4, 2, 2)\n"
              "a = m.logL_st_batch(ss, th, tr)\n"
              "import torch\n"
              "assert torch.cuda.is_available()\n"
              "assert torch.zeros(3, device='cuda').sum().item() == 0\n"
              "assert np.array_equal(m.logL_st_batch(ss, th, tr), a)\n"
              "print('ORDER_OK')\n") % (root, root)
    r = subprocess.run([sys.executable, '-c', script], capture_output=True, text=True, timeout=300)
    assert 'ORDER_OK' in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_baseline_config4_full_samples_per_trajectory(built_lib):
    """
    BASELINE configs[3] at its stated size: 3-state model, T = 2000, 5000 samples PER TRAJECTORY, mixed missing-frame
    masks (none / iid 10 % / bursty 30 %, frame 0 missing in half of the trajectories) -- six trajectories, 30 000
    evaluations in one launch per path; 48 spot checks against the oracle plus batch-placement invariance.
    """
    import bild_amd
    from bild_amd.profiles import segments_from_st
    rng = np.random.default_rng(4404)
    T, per, k = 2000, 5000, 5
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, looppositions=H.LOOPS[3], localization_error=0.1)
    trajs = []
    for j, kind in enumerate(['none', 'iid', 'bursty', 'none', 'iid', 'bursty']):
        miss = H.missing_mask(rng, T, kind)
        if j % 2 == 1:
            miss = np.union1d(miss, [0])
        trajs.append(model.trajectory_from_loopingprofile(H.random_profile(rng, T, 3, 300), missing_frames=miss, rng=rng))
    ss, thetas = H.candidate_profiles(rng, len(trajs) * per, k, 3)
    seg_start, seg_state = segments_from_st(ss, thetas, T)
    tid = np.repeat(np.arange(len(trajs)), per).astype(np.int32)
    results = {}
    for path in ('auto', 'dense'):
        model.path = path
        got = model.logL_segments(seg_start, seg_state, trajs, tid)
        assert got.shape == (len(trajs) * per,) and np.all(np.isfinite(got))
        worst = _spot_check(model, trajs, seg_start, seg_state, tid, got, np.random.default_rng(1), 48, [T] * len(trajs))
        assert worst < TOL, (path, worst)
        results[path] = got
    assert np.max(np.abs(results['auto'] - results['dense'])) < TOL
    # placement inside a batch, the batch's size and its order do not matter, bit for bit -- on the same trajectory set
    model.path = 'auto'
    for j in (1, 5):
        sl = slice(j * per, (j + 1) * per)
        alone = model.logL_segments(seg_start[sl][::-1], seg_state[sl][::-1], trajs, tid[sl][::-1])[::-1]
        assert np.array_equal(alone, results['auto'][sl])
    # one trajectory at a time through the sampler seam: a trajectory set of its own.  Whether a set gets the pair table
    # (api.cpp: ensure_pairs) depends on its size; where two sets differ in that, the pieces of a sum are cut differently and
    # the values agree to rounding (1e-11 observed), not to the bit -- the bar here covers both cases
    for j in (1, 5):
        sl = slice(j * per, (j + 1) * per)
        sampler = bild_amd.FixedkSampler(trajs[j], model, k=k, N=per, max_fcomplete=0)
        assert np.max(np.abs(sampler.logL(ss[sl], thetas[sl]) - results['auto'][sl])) < 1e-9


def test_baseline_config5_full_inference(built_lib):
    """
    BASELINE configs[4] on one GPU: the full adaptive-k loop (`sample_many`, default sampler settings: N = 100,
    init_runs = 20, certainty 0.99) on 64 synthetic trajectories of experimental length, T ~ U{150..600}.
    Against the oracle: >= 32 likelihoods drawn from the samplers' pools, the likelihood of EVERY best profile, and
    the boundary polishing (GPU-driven vs oracle-driven) of every result.
    """
    import bild_amd
    from bild_amd import postproc
    from oracle import oracle
    rng = np.random.default_rng(64)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    trajs = []
    for _ in range(64):
        T = int(rng.integers(150, 601))
        trajs.append(model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 120), rng=rng))
    np.random.seed(640)
    results = bild_amd.sample_many(trajs, model, return_exceptions=True)
    failed = [r for r in results if isinstance(r, Exception)]
    assert not failed, repr(failed[:1])
    oracle_model = _OracleRouse(model)
    n_steps = sum(len(s.samples) for r in results for s in r.samplers)
    assert n_steps > 64 * 20                                  # at least the initial runs of k = 0 everywhere

    # (i) likelihoods out of the samplers' pools
    pools = [(r, s, smp) for r in results for s in r.samplers for smp in s.samples if len(smp['logLs'])]
    worst = 0.0
    for idx in rng.choice(len(pools), 48, replace=False):
        r, s, smp = pools[idx]
        i = int(rng.integers(len(smp['logLs'])))
        T = len(r.traj)
        states = H.expand(smp['ss'][i:i + 1], smp['thetas'][i:i + 1], T)[0]
        want = oracle.logl(model.arrays(), model.measurement, model._get_noise(r.traj), r.traj[:], states)
        worst = max(worst, abs(smp['logLs'][i] - want))
    assert worst < TOL, worst

    # (ii) every best profile, (iii) boundary polishing driven by the GPU and by the oracle
    moved = 0
    for r in results:
        prof = r.best_profile()
        assert abs(model.logL(prof, r.traj) - oracle_model.logL(prof, r.traj)) < TOL
        try:
            pg = postproc.optimize_boundary(prof, r.traj, model)
        except postproc.BoundaryEliminationError:
            with pytest.raises(postproc.BoundaryEliminationError):
                postproc.optimize_boundary(prof, r.traj, oracle_model)
            continue
        pc = postproc.optimize_boundary(prof, r.traj, oracle_model)
        assert pg == pc
        assert model.logL(pg, r.traj) >= model.logL(prof, r.traj) - 1e-9
        moved += int(np.sum(pg[:] != prof[:]))
    # the inference finds the switches it should: most trajectories of 150-600 frames with dwell ~120 have some
    ks = np.array([int(r.best_k()) for r in results])
    assert np.mean(ks > 0) > 0.5, ks


def test_nonsymmetric_propagator_follows_the_textbook_filter(built_lib):
    """
    A propagator B that is not symmetric cannot come out of the reference's model classes (rouse builds B = exp(-kA)
    from a symmetric connectivity matrix); it can only be passed in through `from_arrays`.  On such input the
    reference's two kernels already disagree with each other -- the Cython kernel reads the lower triangle as if B were
    symmetric (dsymv, pyx:210-239), the NumPy kernel forms B C B (MSRouse_logL_py.py:109-110) -- and this library
    follows NEITHER: it runs the Kalman predict  M <- B M + G,  C <- B C B^T + Sig  (dense vector path; the modal
    path and the matrix-pipe kernel refuse the model).  Stated here so that nobody has to guess.
    """
    import bild_amd
    from bild_amd import _lib
    from oracle import oracle
    rng = np.random.default_rng(12)
    N, d, T = 6, 2, 60
    a = {k2: v.copy() for k2, v in H.DuckModel(N=N, d=d).arrays().items()}
    a['B'][0, 0, 1] += 0.02
    a['B'][1, 3, 2] -= 0.015
    w, err = H.end2end(N), np.array([0.2, 0.2])
    model = bild_amd.MultiStateRouse.from_arrays(a['B'], a['G'], a['Sig'], a['M0'], a['C0'], w, localization_error=err)
    assert model.handle().query(_lib.Q_MODAL_OK) == 0
    x = rng.standard_normal((T, d))
    x[7] = np.nan
    states = H.random_profile(rng, T, 2, 15)

    def textbook(states):
        M, C = a['M0'][states[0]].copy(), a['C0'][states[0]].copy()
        tot = 0.0
        for t in range(T):
            if t > 0:
                B, G, Sig = a['B'][states[t]], a['G'][states[t]], a['Sig'][states[t]]
                M, C = B @ M + G, B @ C @ B.T + Sig
            if np.any(np.isnan(x[t])):
                continue
            Cw = C @ w
            S = w @ Cw + err[0] ** 2
            nu = x[t] - w @ M
            tot += np.sum(-0.5 * (nu ** 2 / S + np.log(S) + np.log(2 * np.pi)))
            M = M + np.outer(Cw, nu) / S
            C = C - np.outer(Cw, Cw) / S
        return tot

    got = model.logL(H.ProfileView(states), x)
    assert abs(got - textbook(states)) < TOL
    for flavor in ('numpy', 'cython'):
        other = oracle.logl(a, w, err, x, states, flavor=flavor)
        assert abs(got - other) > 1e-6, flavor        # neither reference kernel: see the docstring
    with pytest.raises(_lib.BildAmdError):
        model.path = 'modal'
        model.logL(H.ProfileView(states), x)


def test_st_seam_against_reference_vectors(built_lib):
    """
    Row a2 on the GPU path: for every batch of tests/golden/st2profile.npz (profiles returned by the REFERENCE's own
    FixedkSampler.st2profile) the seam `logL_st_batch(ss, thetas)` -- native conversion inside bild_logl_st -- must give
    exactly what the expanded reference profiles give through `logL_batch`: any disagreement about a switch frame shows.
    """
    import bild_amd
    from test_profiles import _st_golden
    rng = np.random.default_rng(77)
    models = {S: bild_amd.MultiStateRouse(20, 1, 5, d=3, looppositions=H.LOOPS[S], localization_error=0.1) for S in (2, 3)}
    trajs = {}
    n = 0
    for ss, thetas, T, S, want_states in _st_golden():
        model = models[S]
        if (S, T) not in trajs:
            trajs[S, T] = model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, max(T // 4, 1)), rng=rng)
        a = model.logL_st_batch(ss, thetas, trajs[S, T])
        b = model.logL_batch(want_states, trajs[S, T])
        assert np.array_equal(a, b), (T, S)
        n += len(a)
    assert n > 1000


def test_device_entry_rejects_bad_descriptors_when_asked(built_lib):
    """
    `bild_logl_segments_device` trusts device-resident descriptors unless BILD_VALIDATE_DEVICE is set; with the flag a
    state >= S, a decreasing start, a first start != 0 or a trajectory id out of range comes back as an error instead of
    driving an out-of-range access (nothing is launched for a rejected batch).
    """
    import torch
    import bild_amd
    from bild_amd import _lib
    from bild_amd.profiles import segments_from_st
    rng = np.random.default_rng(21)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    trajs = [model.trajectory_from_loopingprofile(H.random_profile(rng, 120, 2, 30), rng=rng) for _ in range(3)]
    ss, thetas = H.candidate_profiles(rng, 500, 3, 2)
    a, b = segments_from_st(ss, thetas, 120)
    tid = rng.integers(3, size=500).astype(np.int32)
    h, ts = model.handle(), model.trajset(trajs)
    want = model.logL_segments(a, b, trajs, tid)
    dev = torch.device('cuda', 0)
    d_out = torch.full((500,), 7.0, dtype=torch.float64, device=dev)

    def run(a_, b_, tid_, validate=True):
        da, db, dt_ = (torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (a_, b_, tid_))
        _lib.logl_segments_device(h, ts, 500, 4, da.data_ptr(), db.data_ptr(), dt_.data_ptr(), d_out.data_ptr(),
                                  stream=torch.cuda.current_stream().cuda_stream, validate=validate)
        torch.cuda.synchronize()
        return d_out.cpu().numpy()

    assert np.array_equal(run(a, b, tid), want)
    assert np.array_equal(run(a, b, tid, validate=False), want)
    for what, (a2, b2, t2) in {
        'state': (a, np.where(np.arange(500)[:, None] == 17, 2, b).astype(np.int32), tid),
        'negative state': (a, np.where(np.arange(500)[:, None] == 3, -1, b).astype(np.int32), tid),
        'order': (np.where((np.arange(500)[:, None] == 400) & (np.arange(4)[None, :] == 2), 0, a).astype(np.int32), b, tid),
        'first start': (np.where((np.arange(500)[:, None] == 9) & (np.arange(4)[None, :] == 0), 1, a).astype(np.int32), b, tid),
        'traj_id': (a, b, np.where(np.arange(500) == 250, 3, tid).astype(np.int32)),
    }.items():
        d_out.fill_(7.0)
        with pytest.raises(_lib.BildAmdError) as info:
            run(a2, b2, t2)
        assert info.value.code == _lib.ERR_INVALID, what
        assert np.all(d_out.cpu().numpy() == 7.0), what            # nothing ran


def test_one_model_on_two_streams_with_distinct_errors(built_lib):
    """
    d* = 2 (localization errors that differ between dimensions) needs a buffer of partial results per launch; it belongs
    to the call, so launches of ONE model on different streams, interleaved with host-entry calls, do not disturb each other.
    """
    import torch
    import bild_amd
    from bild_amd import _lib
    from bild_amd.profiles import segments_from_st
    rng = np.random.default_rng(22)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=[0.1, 0.1, 0.3])
    T = 400
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 80), rng=rng)
    h, ts = model.handle(), model.trajset(traj)
    dev = torch.device('cuda', 0)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    batches = []
    for i, n in enumerate((6000, 5000)):
        ss, thetas = H.candidate_profiles(rng, n, 4, 2)
        a, b = segments_from_st(ss, thetas, T)
        batches.append((n, torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev),
                        torch.empty(n, dtype=torch.float64, device=dev), model.logL_segments(a, b, traj), ss, thetas))
    torch.cuda.synchronize()
    for rep in range(4):
        for (n, da, db, dout, want, ss, thetas), st in zip(batches, streams):
            _lib.logl_segments_device(h, ts, n, 5, da.data_ptr(), db.data_ptr(), 0, dout.data_ptr(), stream=st.cuda_stream)
        host = model.logL_st_batch(batches[0][5][:700], batches[0][6][:700], traj)   # host entry in between
        torch.cuda.synchronize()
        assert np.array_equal(host, batches[0][4][:700])
        for n, da, db, dout, want, ss, thetas in batches:
            assert np.array_equal(dout.cpu().numpy(), want)


@pytest.mark.parametrize('case', ['one_traj', 'many_traj_dstar2_missing', 'three_state'])
def test_prefix_table_and_launch_order_do_not_change_results(built_lib, case):
    """
    Candidates start from the prefix table (state of the switch-free recursion in front of their first switch) and are
    dealt to wavefronts in an order chosen by the host scheduler.  Both are matters of speed: every result must be
    bit-identical to the run from frame 0 in array order (BILD_NO_PREFIX), through the host and the device entry points,
    and equal to the oracle.  Includes profiles that never switch, first switches at frame 1, empty later segments.
    """
    import torch
    import bild_amd
    from bild_amd import _lib
    from bild_amd.profiles import segments_from_st
    rng = np.random.default_rng({'one_traj': 1, 'many_traj_dstar2_missing': 2, 'three_state': 3}[case])
    if case == 'one_traj':
        model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
        Ts, n, k, S = [700], 9000, 4, 2
        miss = [None]
    elif case == 'many_traj_dstar2_missing':
        model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=[0.1, 0.25, 0.1])
        Ts, n, k, S = [150, 333, 90, 611, 12], 7000, 3, 2
        miss = [None, 0.1, [0, 1, 5], 0.3, None]
    else:
        model = bild_amd.MultiStateRouse(24, 1, 4, d=2, looppositions=H.LOOPS[3], localization_error=0.15)
        Ts, n, k, S = [400, 250], 5000, 6, 3
        miss = [0.05, None]
    trajs = [model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, max(T // 4, 2)), missing_frames=m_, rng=rng)
             for T, m_ in zip(Ts, miss)]
    tid = rng.integers(len(trajs), size=n).astype(np.int32)
    ss, thetas = H.candidate_profiles(rng, n, k, S)
    seg_start = np.zeros((n, k + 1), dtype=np.int32)
    for j, T in enumerate(Ts):
        sel = tid == j
        seg_start[sel] = segments_from_st(ss[sel], thetas[sel], T)[0]
    seg_state = thetas.astype(np.int32)
    seg_start[:40, 1:] = np.int32(2 ** 31 - 1)       # never switch
    seg_start[40:80, 1] = 1                          # first switch at frame 1
    seg_start[80:120, 2] = seg_start[80:120, 1]      # an empty segment behind the first switch
    seg_start[:, 1:] = np.sort(seg_start[:, 1:], axis=1)
    h, ts = model.handle(), model.trajset(trajs)
    base = _lib.logl_segments(h, ts, seg_start, seg_state, tid, prefix=False)
    exact = _lib.logl_segments(h, ts, seg_start, seg_state, tid, jump=False)     # table as starting point only
    fast = _lib.logl_segments(h, ts, seg_start, seg_state, tid)                  # + convergence jumps (the default)
    assert _lib.prefix_info(ts)[0] > 0                # a table was built
    assert np.array_equal(exact, base)
    # the default launch is split (table walk + frame loop over the work lists, csrc/walk.hip): same numbers, same order
    assert np.array_equal(_lib.logl_segments(h, ts, seg_start, seg_state, tid, split=False), fast)
    # ... and chains of close switches start at their second switch, from the transient state table: the same numbers again
    assert np.array_equal(_lib.logl_segments(h, ts, seg_start, seg_state, tid, states=False), fast)
    assert np.array_equal(_lib.logl_segments(h, ts, seg_start, seg_state, tid, states=False, split=False), fast)
    jump_dev = np.max(np.abs(fast - base))
    print(f"{case}: max |jumping - frame by frame| = {jump_dev:.2e} on |logL| up to {np.max(np.abs(base)):.1e}")
    assert jump_dev < 1e-9
    # device entry: array order and scheduled order, with and without the table / the jumps
    dev = torch.device('cuda', 0)
    order = _lib.schedule_segments(h, ts, seg_start, seg_state, tid)
    assert np.array_equal(np.sort(order), np.arange(n))
    d = {name: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for name, v in
         dict(a=seg_start, b=seg_state, t=tid, o=order).items()}
    out = torch.empty(n, dtype=torch.float64, device=dev)
    for d_order in (0, d['o'].data_ptr()):
        for prefix, jump, split, want in ((True, False, True, base), (False, True, True, base), (True, True, True, fast),
                                          (True, True, False, fast)):
            out.fill_(0.0)
            _lib.logl_segments_device(h, ts, n, k + 1, d['a'].data_ptr(), d['b'].data_ptr(), d['t'].data_ptr(), out.data_ptr(),
                                      stream=torch.cuda.current_stream().cuda_stream, d_order=d_order, prefix=prefix, jump=jump,
                                      split=split, validate=True)
            torch.cuda.synchronize()
            # a result depends on its own candidate only: never on the launch order or on the rest of the batch
            assert np.array_equal(out.cpu().numpy(), want), (d_order != 0, prefix, jump, split)
    half = rng.permutation(n)[:n // 3]
    assert np.array_equal(_lib.logl_segments(h, ts, seg_start[half], seg_state[half], tid[half]), fast[half])
    # a launch order that is not a permutation is refused when validation is asked for
    bad = order.copy()
    bad[5] = bad[6]
    d_bad = torch.from_numpy(bad).to(dev)
    with pytest.raises(_lib.BildAmdError):
        _lib.logl_segments_device(h, ts, n, k + 1, d['a'].data_ptr(), d['b'].data_ptr(), d['t'].data_ptr(), out.data_ptr(),
                                  stream=torch.cuda.current_stream().cuda_stream, d_order=d_bad.data_ptr(), validate=True)
    # segment 0 owns frame 0: a later segment may not start there
    zero = seg_start.copy()
    zero[3, 1] = 0
    with pytest.raises(_lib.BildAmdError):
        _lib.logl_segments(h, ts, zero, seg_state, tid)
    # and against the oracle (the special rows above all among the checked ones)
    pick = np.concatenate([np.arange(0, 120, 6), rng.choice(n, 30, replace=False)])
    assert _spot_check(model, trajs, seg_start[pick], seg_state[pick], tid[pick], fast[pick], rng, len(pick), Ts) < TOL
    # frames the tasks ran themselves (counted on the device): all of them without the table, a fraction with it
    total = float(np.sum(np.asarray(Ts)[tid])) * (2 if case == 'many_traj_dstar2_missing' else 1)
    shares = {}
    for name, kw in (('frame by frame', dict(prefix=False)), ('table', dict(jump=False)), ('table + jumps', {})):
        _lib.kernel_timing(True)
        _lib.logl_segments(h, ts, seg_start, seg_state, tid, **kw)
        _lib.kernel_timing(False)
        _lib.kernel_timing_read()
        shares[name] = _lib.frames_run_read(h) / total
    print(case, {k_: round(v, 3) for k_, v in shares.items()})
    assert 0.97 < shares['frame by frame'] <= 1.0          # frame 0 is not counted
    assert shares['table + jumps'] < shares['table'] < shares['frame by frame']


def test_launch_order_of_a_throughput_bound_batch(built_lib):
    """
    Batches of several rounds with more busy candidates than one round has waves get the sorted launch order (candidates of
    equal work share a wave), smaller ones the spread order (api.cpp: schedule).  Either way a permutation, and never a
    different result: 30 000 candidates with six switches each on a short trajectory, against the same batch in array order.
    """
    import torch
    import bild_amd
    from bild_amd import _lib
    from bild_amd.profiles import segments_from_st
    rng = np.random.default_rng(31)
    T, n, k = 300, 30000, 6
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 60), rng=rng)
    ss, thetas = H.candidate_profiles(rng, n, k, 2)
    a, b = segments_from_st(ss, thetas, T)
    h, ts = model.handle(), model.trajset(traj)
    want = _lib.logl_segments(h, ts, a, b, None)                       # host entry, first evaluation on the set: array order
    # ... from the second evaluation on the host entries order batches of several rounds on the device (schedule.hip)
    assert np.array_equal(_lib.logl_segments(h, ts, a, b, None), want)
    assert np.array_equal(_lib.logl_segments(h, ts, a[::-1].copy(), b[::-1].copy(), None)[::-1], want)
    dev = torch.device('cuda', 0)
    da, db = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    for count in (n, 9000):                                             # several rounds / one round
        order = _lib.schedule_segments(h, ts, a[:count], b[:count])
        assert np.array_equal(np.sort(order), np.arange(count))
        d_order = torch.from_numpy(order).to(dev)
        out.fill_(0.0)
        _lib.logl_segments_device(h, ts, count, k + 1, da.data_ptr(), db.data_ptr(), 0, out.data_ptr(), d_order=d_order.data_ptr(),
                                  validate=True)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy()[:count], want[:count])
    pick = rng.choice(n, 24, replace=False)
    assert _spot_check(model, [traj], a[pick], b[pick], np.zeros(len(pick), np.int32), want[pick], rng, len(pick), [T]) < TOL


@pytest.mark.parametrize('S', [2, 3])
def test_pair_table_two_close_switches_run_no_frame(built_lib, S):
    """
    Second-level transient table (csrc/common.h "pair table", api.cpp ensure_pairs): two switches closer together than the
    first one's transient come out of ONE entry, keyed by (old, middle, new state, frame, gap).  Designed candidates:
    pairs at every kind of place (frame 1, mid-trajectory, the second switch on the last frame / beyond the end), gaps
    from 1 up to beyond the table's range, followed by nothing, by a far third switch, or by a near one (a chain of
    three, which is run).  Results equal the frame-by-frame run and the oracle; the pairs inside the table's range run
    no frame at all (counted on the device).
    """
    import ctypes
    import torch
    import bild_amd
    from bild_amd import _lib
    rng = np.random.default_rng(77 + S)
    T = 500
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, looppositions=H.LOOPS[S], localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, 100), missing_frames=0.04, rng=rng)
    BIG = np.int32(2 ** 31 - 1)
    rows, kinds = [], []
    for t in (1, 2, 37, 250, 430, 470, 495, 498):
        for g in (1, 2, 5, 11, 20, 33, 47, 63, 64, 65, 90):
            for tail in ('none', 'far', 'near'):
                t2 = t + g
                t3 = {'none': BIG, 'far': t2 + 150, 'near': t2 + 6}[tail]
                rows.append([0, t, t2, t3])
                kinds.append((t, g, tail))
    seg_start = np.array(rows, dtype=np.int32)
    n = len(seg_start)
    seg_state = np.zeros((n, 4), dtype=np.int32)
    seg_state[:, 0] = rng.integers(S, size=n)
    for i in range(1, 4):                                   # every boundary a real switch; S = 3: both A-B-A and A-B-C
        step = rng.integers(1, S, size=n)
        seg_state[:, i] = (seg_state[:, i - 1] + step) % S
    h, ts = model.handle(), model.trajset(traj)
    base = _lib.logl_segments(h, ts, seg_start, seg_state, None, prefix=False)
    fast = _lib.logl_segments(h, ts, seg_start, seg_state, None)
    print(f"S={S}: max |tables - frame by frame| = {np.max(np.abs(fast - base)):.2e}")
    assert np.max(np.abs(fast - base)) < 1e-9
    pick = rng.choice(n, 40, replace=False)
    assert _spot_check(model, [traj], seg_start[pick], seg_state[pick], np.zeros(len(pick), np.int32), fast[pick], rng, len(pick), [T]) < TOL
    # frames run per candidate
    dev = torch.device('cuda', 0)
    da, db = torch.from_numpy(seg_start).to(dev), torch.from_numpy(seg_state).to(dev)
    out = torch.empty(n, dtype=torch.float64, device=dev)
    frames = torch.full((n,), -1, dtype=torch.int32, device=dev)
    _lib.lib().bild_debug_frames_per_task(ctypes.c_void_p(frames.data_ptr()))
    try:
        _lib.logl_segments_device(h, ts, n, 4, da.data_ptr(), db.data_ptr(), 0, out.data_ptr())
        torch.cuda.synchronize()
    finally:
        _lib.lib().bild_debug_frames_per_task(None)
    assert np.array_equal(out.cpu().numpy(), fast)
    f = frames.cpu().numpy()
    in_table = near = 0
    for i, (t, g, tail) in enumerate(kinds):
        t2, third_inside = t + g, int(seg_start[i, 3]) < T
        if t2 >= T:
            assert f[i] == 0, (kinds[i], f[i])               # a lone switch in front of the end: single table
        elif g <= 20 and not (tail == 'near' and third_inside):
            assert f[i] == 0, (kinds[i], f[i])               # the pair -- converged before the far third switch, or reaching
            in_table += 1                                    # the end of the trajectory: one entry of the pair table
        elif g <= 20:
            assert f[i] > 0, (kinds[i], f[i])                # three close switches: run
            near += 1
    assert in_table >= 40 and near >= 20
    assert _lib.prefix_info(ts)[0] > 0


@pytest.mark.parametrize('d,err', [(4, 0.1), (5, [0.1, 0.1, 0.1, 0.1, 0.3]), (6, [0.2, 0.1, 0.2, 0.1, 0.2, 0.1]), (8, 0.15),
                                   (7, [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7])])
@pytest.mark.parametrize('path', ['auto', 'dense'])
def test_more_than_three_dimensions(built_lib, d, err, path):
    """
    The reference takes any d (bild/models.py:222).  A task carries up to three mean vectors; dimensions beyond that --
    with one localization error or several -- run as further covariance chains of the same sample (d <= 8).
    """
    import bild_amd
    from oracle import oracle
    rng = np.random.default_rng(10 * d + len(path))
    T, n, k = 160, 150, 3
    model = bild_amd.MultiStateRouse(16, 1, 4, d=d, localization_error=err, path=path)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 40), missing_frames=0.06, rng=rng)
    ss, thetas = H.candidate_profiles(rng, n, k, 2)
    got = model.logL_st_batch(ss, thetas, traj)
    want = oracle.logl_batch(model.arrays(), model.measurement, model.localization_error, traj[:], H.expand(ss, thetas, T))
    assert np.max(np.abs(got - want)) < TOL
    one = model.logL(H.ProfileView(H.expand(ss[:1], thetas[:1], T)[0]), traj)
    assert abs(one - want[0]) < TOL


@pytest.mark.parametrize('S', [6, 9, 24, 48])
def test_many_states(built_lib, S):
    """
    The reference takes any number of states (bild/models.py:242-247).  Up to the point where all S*S basis changes fit
    a quarter of LDS they are kept as pairs; beyond, as the 2 S factors Q[s], Q[s]^T and a switch changes basis in two
    steps (S = 9 at 12 modes is already there).  Against the oracle, on both paths the model admits.
    """
    import bild_amd
    from bild_amd import _lib
    from oracle import oracle
    rng = np.random.default_rng(S)
    N, T, n, k = 12, 150, 200, 5
    pairs = [(a, b) for span in range(2, N) for a in range(N - span) for b in [a + span]]        # 55 distinct bonds
    loops = [None] + [pairs[i % len(pairs)] + (1.0 + 0.5 * (i // len(pairs)),) for i in range(S - 1)]
    model = bild_amd.MultiStateRouse(N, 1, 3, d=2, looppositions=tuple(loops), localization_error=0.1)
    assert model.nStates == S
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, 30), missing_frames=0.05, rng=rng)
    ss, thetas = H.candidate_profiles(rng, n, k, S)
    want = oracle.logl_batch(model.arrays(), model.measurement, model.localization_error, traj[:], H.expand(ss, thetas, T))
    for path in (['auto', 'dense'] if S <= 24 else ['auto']):
        model.path = path
        got = model.logL_st_batch(ss, thetas, traj)
        assert np.max(np.abs(got - want)) < TOL, (S, path)
        exact = _lib.logl_st(model.handle(), model.trajset(traj), ss, thetas, path=path, prefix=False)
        assert np.max(np.abs(exact - want)) < TOL
    if S == 48:
        # the envelope: beyond ~50 states at 12 modes even the factors no longer fit LDS -- refused with a message
        big = bild_amd.MultiStateRouse(N, 1, 3, d=2, looppositions=tuple([None] + [pairs[i % 55] + (1.0 + i,) for i in range(119)]),
                                       localization_error=0.1)
        with pytest.raises(_lib.BildAmdError) as info:
            big.logL_st_batch(ss[:4], thetas[:4] % 2, traj)
        assert info.value.code == _lib.ERR_UNSUPPORTED and 'too many states' in str(info.value)


def test_st_rows_resident_in_hbm(built_lib):
    """
    bild_logl_st_device: the sampler's (s, theta) rows lie in HBM (float64 / uint8) and everything happens on the device --
    switch frames as st2profile computes them, cleaning, table walk, frame loop.  Bit-identical to the host-buffer seam
    (`bild_logl_st`), to the segment entry on the host-converted lists, on the caller's stream and on a second one, for one
    trajectory and for several (traj_id, two localization-error chains); a row that is no point on the simplex gets NaN and
    is reported in the status word, the other rows are unaffected.
    """
    import torch
    import bild_amd
    from bild_amd import _lib
    from bild_amd.profiles import segments_from_st
    rng = np.random.default_rng(321)
    dev = torch.device('cuda', 0)
    for case in ('one', 'many_dstar2'):
        if case == 'one':
            model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
            Ts, n, k = [600], 6000, 5
        else:
            model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=[0.1, 0.25, 0.1])
            Ts, n, k = [150, 333, 611], 5000, 3
        trajs = [model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, max(T // 4, 2)), missing_frames=0.05, rng=rng) for T in Ts]
        tid = rng.integers(len(trajs), size=n).astype(np.int32)
        ss, thetas = H.candidate_profiles(rng, n, k, 2)
        h, ts = model.handle(), model.trajset(trajs)
        want = _lib.logl_st(h, ts, ss, thetas, tid)
        a = np.zeros((n, k + 1), dtype=np.int32)
        for j, T in enumerate(Ts):
            a[tid == j] = segments_from_st(ss[tid == j], thetas[tid == j], T)[0]
        assert np.array_equal(_lib.logl_segments(h, ts, a, thetas.astype(np.int32), tid), want)
        d_ss = torch.from_numpy(ss).to(dev)
        d_th = torch.from_numpy(thetas.astype(np.uint8)).to(dev)
        d_tid = torch.from_numpy(tid).to(dev)
        out = torch.zeros(n, dtype=torch.float64, device=dev)
        status = torch.zeros(2, dtype=torch.int32, device=dev)
        side = torch.cuda.Stream()
        for stream in (torch.cuda.current_stream(), side, torch.cuda.current_stream()):
            for split in (True, False):
                out.fill_(0.0)
                torch.cuda.synchronize()
                _lib.logl_st_device(h, ts, n, k + 1, d_ss.data_ptr(), d_th.data_ptr(), out.data_ptr(), d_traj_id=d_tid.data_ptr(),
                                    stream=stream.cuda_stream, d_status=status.data_ptr(), split=split)
                torch.cuda.synchronize()
                assert np.array_equal(out.cpu().numpy(), want), (case, split)
        assert status.cpu().numpy()[0] == 0
        bad = ss.copy()
        bad[11, 1] = np.nan
        bad[12, 0] = -0.3
        d_bad = torch.from_numpy(bad).to(dev)
        _lib.logl_st_device(h, ts, n, k + 1, d_bad.data_ptr(), d_th.data_ptr(), out.data_ptr(), d_traj_id=d_tid.data_ptr(),
                            d_status=status.data_ptr())
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.isnan(got[11]) and np.isnan(got[12]) and status.cpu().numpy()[0] == 1 and status.cpu().numpy()[1] in (11, 12)
        keep = np.ones(n, dtype=bool)
        keep[[11, 12]] = False
        assert np.array_equal(got[keep], want[keep])
        with pytest.raises(_lib.BildAmdError):
            _lib.logl_st(h, ts, bad, thetas, tid)


def test_schedule_rejects_rows_it_cannot_index(built_lib):
    """ bild_schedule_segments validates like bild_logl_segments: a decreasing row used to index its work histogram out of range """
    import bild_amd
    from bild_amd import _lib
    rng = np.random.default_rng(2)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, 1000, 2, 200), rng=rng)
    h, ts = model.handle(), model.trajset(traj)
    a = np.array([[0, 500, 100], [0, 10, 20]], dtype=np.int32)
    b = np.array([[0, 1, 0], [0, 1, 0]], dtype=np.int32)
    _lib.logl_segments(h, ts, a[1:], b[1:])            # the set's tables exist now
    with pytest.raises(_lib.BildAmdError):
        _lib.schedule_segments(h, ts, a, b)
    with pytest.raises(_lib.BildAmdError):
        _lib.schedule_segments(h, ts, a[1:], np.array([[0, 7, 0]], dtype=np.int32))
    assert len(_lib.schedule_segments(h, ts, a[1:], b[1:])) == 1


def test_trajset_cache_notices_edits_in_place(built_lib):
    """
    The trajectory-set cache is keyed by identity and guarded by a look at the contents (address, shape, the sum of the
    bit patterns of the values): masking frames or rescaling the data in place leads to a fresh upload instead of stale
    likelihoods; a trajectory-like that builds a new array on every access is keyed by its contents and uploaded once.
    """
    import bild_amd
    rng = np.random.default_rng(4)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, 300, 2, 60), rng=rng)
    prof = H.random_profile(rng, 300, 2, 60)
    first = model.logL(prof, traj)
    ts0 = model.trajset(traj)
    assert model.trajset(traj) is ts0                       # unchanged: the same resident set
    data = traj[:]
    data[40:50] = np.nan                                      # in place: masked frames
    masked = model.logL(prof, traj)
    assert model.trajset(traj) is not ts0 and masked != first
    fresh = bild_amd.Trajectory(data.copy(), localization_error=traj.localization_error)
    assert masked == model.logL(prof, fresh)
    before = model.trajset(traj)
    data[123, 1] += 1e-9                                      # in place: ONE value, by a hair
    assert model.trajset(traj) is not before
    data *= 1.5                                               # in place: rescaled
    assert model.logL(prof, traj) == model.logL(prof, bild_amd.Trajectory(data.copy(), localization_error=traj.localization_error))

    class Fresh:                                              # t[:] materialises a new array on every access
        def __init__(self, arr):
            self._a, self.localization_error = arr, None
        def __len__(self):
            return len(self._a)
        def __getitem__(self, key):
            return self._a.copy()[key]
    lazy = Fresh(data.copy())
    v0 = model.logL(prof, lazy)
    ts1 = model.trajset(lazy)
    assert model.trajset(lazy) is ts1 and model.logL(prof, lazy) == v0


def test_to_device_call_followed_by_a_host_call(built_lib):
    """
    bild_logl_st_to_device waits for nothing: its kernels still read the model's staging blocks and work lists when the call
    returns.  The next call on the model -- here a host-buffer call with DIFFERENT candidates, right behind it -- must not
    overwrite them early (the event of the first call is recorded behind its last kernel).  Both results against the
    oracle; and a row that is no point on the simplex in a to_device call is reported by bild_logl_st_status.
    """
    import torch
    import bild_amd
    from bild_amd import _lib
    rng = np.random.default_rng(808)
    T, n, k = 800, 20000, 6
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 150), rng=rng)
    h, ts = model.handle(), model.trajset(traj)
    ss1, th1 = H.candidate_profiles(rng, n, k, 2)
    ss2, th2 = H.candidate_profiles(rng, n, k, 2)
    want1, want2 = _lib.logl_st(h, ts, ss1, th1), _lib.logl_st(h, ts, ss2, th2)
    dev = torch.device('cuda', 0)
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    side = torch.cuda.Stream()
    for _ in range(5):
        out.fill_(0.0)
        torch.cuda.synchronize()
        _lib.logl_st_to_device(h, ts, ss1, th1, out.data_ptr(), stream=side.cuda_stream)
        got2 = _lib.logl_st(h, ts, ss2, th2)                 # immediately: other candidates through the same staging blocks
        torch.cuda.synchronize()
        assert np.array_equal(got2, want2) and np.array_equal(out.cpu().numpy(), want1)
    _lib.logl_st_status(h)                                   # nothing was refused
    bad = ss1.copy()
    bad[5, 2] = -1.0
    _lib.logl_st_to_device(h, ts, bad, th1, out.data_ptr(), stream=side.cuda_stream)
    with pytest.raises(_lib.BildAmdError):
        _lib.logl_st_status(h)
    _lib.logl_st_status(h)                                   # the verdict is forgotten once it has been read
    assert np.isnan(out.cpu().numpy()[5])
    pick = rng.choice(n, 12, replace=False)
    from oracle import oracle
    ref = oracle.logl_batch(model.arrays(), model.measurement, model.localization_error, traj[:], H.expand(ss1[pick], th1[pick], T))
    assert np.max(np.abs(ref - want1[pick])) < TOL


def test_declared_use_decides_which_tables_are_built(built_lib):
    """
    bild_trajset_expect: a set that will see a handful of evaluations builds no tables (cheaper frame by frame), a few
    hundred build the prefix and transient tables, thousands (or no declaration) all of them.  The results agree to the
    jumps' tolerance whatever was declared, and a declaration after the first evaluation is refused.
    """
    import bild_amd
    from bild_amd import _lib
    rng = np.random.default_rng(17)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    data = model.trajectory_from_loopingprofile(H.random_profile(rng, 500, 2, 100), rng=rng)
    ss, thetas = H.candidate_profiles(rng, 400, 4, 2)
    out, sizes = {}, {}
    for expect in (10, 1000, 10 ** 6, None):
        traj = bild_amd.Trajectory(data[:].copy(), localization_error=data.localization_error)
        ts = model.trajset(traj, expect=expect)
        out[expect] = _lib.logl_st(model.handle(), ts, ss, thetas)
        sizes[expect] = _lib.prefix_info(ts)[0]
        with pytest.raises(_lib.BildAmdError):
            ts.expect(5)
    assert sizes[10] == 0 < sizes[1000] < sizes[10 ** 6] == sizes[None]
    assert np.array_equal(out[10 ** 6], out[None])
    assert np.array_equal(out[10], _lib.logl_st(model.handle(), model.trajset(data), ss, thetas, prefix=False))
    for expect in (1000, None):
        assert np.max(np.abs(out[expect] - out[10])) < 1e-9


@pytest.mark.parametrize('N,S,k,n', [(20, 3, 4, 3000), (20, 3, 8, 3000), (40, 2, 4, 3000), (20, 3, 8, 30000), (20, 2, 12, 20000)])
def test_listed_frame_loop_of_longer_chains(built_lib, N, S, k, n):
    """
    Chains of 16 and 20 modes (the 3-state model of BASELINE configs[3]; a 40-monomer 2-state chain): the frame loop over the
    work lists runs in a geometry of its own (one wave per SIMD, kernels.hip: listed_geometry) when the estimated list is
    short.  Same arithmetic: bit-identical to the single launch and to the launch in the batch geometry; < 1e-8 vs the oracle.
    The two large cases have work lists of several layers and rounds (dealt out in snake order, three or four rows per wave):
    every task must be run exactly once.
    """
    import os
    import bild_amd
    from bild_amd import _lib
    from bild_amd.profiles import segments_from_st
    rng = np.random.default_rng(500 + N + S + k)
    T = 600
    model = bild_amd.MultiStateRouse(N, 1, 5, d=3, looppositions=H.LOOPS[S], localization_error=0.1)
    assert model.handle().query(_lib.Q_NP) in (10, 16, 20)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, 120), missing_frames=0.03, rng=rng)
    ss, th = H.candidate_profiles(rng, n, k, S)
    h, ts = model.handle(), model.trajset(traj)
    for _ in range(2):                                       # (the tables of a set are built at its first evaluations)
        got = _lib.logl_st(h, ts, ss, th)
    single = _lib.logl_st(h, ts, ss, th, split=False)
    os.environ['BILD_NO_LISTED_GEOMETRY'] = '1'
    try:
        batch_geometry = _lib.logl_st(h, ts, ss, th)
    finally:
        del os.environ['BILD_NO_LISTED_GEOMETRY']
    assert np.array_equal(got, single)
    assert np.array_equal(got, batch_geometry)
    a, b = segments_from_st(ss, th, T)
    assert _spot_check(model, [traj], a, b, np.zeros(n, np.int32), got, rng, 12, [T]) < TOL


def test_refused_row_gets_nan_without_tables_too(built_lib):
    """
    Entries that leave their results on the device cannot refuse a row of (s, theta) that is no point on the simplex: it
    gets NaN and the status word.  That must hold for the single launch as well -- a set that is not worth tables
    (`expect`), the first evaluations on any set, BILD_NO_SPLIT -- where every list is converted and run: the refused row
    runs a marked list without a switch and its result is replaced behind the frame loop.
    """
    import torch
    import bild_amd
    from bild_amd import _lib
    rng = np.random.default_rng(4242)
    T, n, k = 300, 500, 3
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 60), rng=rng)
    h = model.handle()
    ts = model.trajset(traj, expect=10)                      # a handful of evaluations: no tables, no split launch
    ss, th = H.candidate_profiles(rng, n, k, 2)
    want = _lib.logl_st(h, ts, ss, th)
    assert _lib.prefix_info(ts)[0] == 0
    bad = ss.copy()
    bad[7, 1] = np.nan
    bad[8, 0] = -0.25
    dev = torch.device('cuda', 0)
    d_ss = torch.from_numpy(bad).to(dev)
    d_th = torch.from_numpy(th.astype(np.uint8)).to(dev)
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    status = torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.logl_st_device(h, ts, n, k + 1, d_ss.data_ptr(), d_th.data_ptr(), out.data_ptr(), d_status=status.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isnan(got[7]) and np.isnan(got[8])
    assert status.cpu().numpy()[0] == 1 and status.cpu().numpy()[1] in (7, 8)
    keep = np.ones(n, dtype=bool)
    keep[[7, 8]] = False
    assert np.array_equal(got[keep], want[keep])
    with pytest.raises(_lib.BildAmdError):
        _lib.logl_st(h, ts, bad, th)                         # the host-buffer entry refuses the batch
