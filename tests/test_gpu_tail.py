"""
The first-order tail of a transient (csrc/tail.hip, kernels.hip: compare_with_table): once a candidate's covariance has
converged onto the switch-free filter's, the remaining deviation of its means is accounted for by a dot product with a
vector kept beside the prefix table, and the rest of the segment comes out of the table -- 20-25 frames earlier than by
waiting for the means.  Against the frame-by-frame run (which is a pure function of its inputs), against the same
launch without tails (BILD_NO_TAIL), and against the oracle; frames run are counted on the device.
"""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _frames(model, fn):
    from bild_amd import _lib
    _lib.kernel_timing(1)
    out = fn()
    _lib.kernel_timing(False)
    _lib.kernel_timing_read()
    _lib.kernel_timing_read_walk()
    return out, _lib.frames_run_read(model.handle())


@pytest.mark.parametrize('N,S,T,k,miss,err', [(20, 2, 1000, 4, 'none', 0.1), (20, 2, 1000, 8, 'none', 0.1), (20, 2, 600, 6, 'iid', 0.1),
                                              (20, 3, 800, 5, 'bursty', [0.1, 0.1, 0.3]), (32, 2, 1000, 4, 'none', 0.1),
                                              (24, 3, 700, 10, 'none', 0.1), (20, 3, 676, 7, 'none', 0.3),
                                              (12, 2, 400, 12, 'none', 0.05), (40, 2, 1500, 3, 'none', 0.1)])
def test_tails_against_frame_by_frame(built_lib, N, S, T, k, miss, err):
    import bild_amd
    from bild_amd import _lib
    from oracle import oracle
    rng = np.random.default_rng(N + T + k)
    model = bild_amd.MultiStateRouse(N, 1, 5, d=3, looppositions=H.LOOPS[S], localization_error=err)
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, T // 5), missing_frames=H.missing_mask(rng, T, miss), rng=rng)
    ss, th = H.candidate_profiles(rng, 6000, k, S)
    h, ts = model.handle(), model.trajset(traj)
    _lib.logl_st(h, ts, ss[:10], th[:10])                        # tables
    with_tail, f_tail = _frames(model, lambda: _lib.logl_st(h, ts, ss, th))
    without, f_plain = _frames(model, lambda: _lib.logl_st(h, ts, ss, th, tail=False))
    exact = _lib.logl_st(h, ts, ss, th, jump=False)
    scale = max(1.0, float(np.max(np.abs(exact))) / 1e4)
    print(f"N={N} S={S} T={T} k={k} {miss}: frames run {f_plain} -> {f_tail} ({f_tail / max(f_plain, 1):.2f}); |tail - frame by frame| "
          f"{np.max(np.abs(with_tail - exact)):.1e}, |no tail - frame by frame| {np.max(np.abs(without - exact)):.1e}")
    assert np.max(np.abs(with_tail - exact)) < 2e-10 * scale
    assert np.max(np.abs(without - exact)) < 2e-10 * scale
    # the tails add nothing to the deviation the tables already have (a fixed margin behind the table's own transient let a
    # chain's last transient of a 3-state model through with 6e-10; the frames asked for now follow the measured deviation)
    assert np.max(np.abs(with_tail - exact)) < np.max(np.abs(without - exact)) + 2e-11 * scale
    assert f_tail <= f_plain                                      # never more frames
    if miss == 'none' and k <= 8:
        assert f_tail < 0.9 * f_plain                             # ... and markedly fewer where chains end in a long segment
    # the tails change nothing about the invariances: single launch == split launch == chains run from their first switch
    # instead of the state table, bit for bit
    assert np.array_equal(with_tail, _lib.logl_st(h, ts, ss, th, split=False))
    a, b = _lib.segments_from_st(ss, th, T, S)
    assert np.array_equal(with_tail, _lib.logl_segments(h, ts, a, b, states=False))
    assert np.array_equal(with_tail, _lib.logl_segments(h, ts, a, b, states=False, split=False))
    pick = rng.choice(len(ss), 40, replace=False)
    want = oracle.logl_batch(model.arrays(), model.measurement, model.localization_error, traj[:], H.expand(ss[pick], th[pick], T))
    assert np.max(np.abs(with_tail[pick] - want)) < 1e-8


def test_tails_on_data_the_model_did_not_produce(built_lib):
    """ an offset of 1e3 end-to-end distances and 50-sigma outliers: the first-order term scales with the innovations """
    import bild_amd
    from bild_amd import _lib
    rng = np.random.default_rng(77)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    T = 800
    base = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 150), rng=rng)[:]
    for name, x in (('offset', base + 1e3), ('outliers', np.where(rng.random((T, 1)) < 0.01, base + 50 * 0.1 * rng.standard_normal((T, 3)), base))):
        traj = bild_amd.Trajectory(x)
        ss, th = H.candidate_profiles(rng, 4000, 5, 2)
        h, ts = model.handle(), model.trajset(traj)
        got = _lib.logl_st(h, ts, ss, th)
        exact = _lib.logl_st(h, ts, ss, th, jump=False)
        rel = np.max(np.abs(got - exact)) / max(np.max(np.abs(exact)), 1e4)
        print(f"{name}: |logL| ~ {np.max(np.abs(exact)):.1e}, tails vs frame by frame {np.max(np.abs(got - exact)):.1e} (relative {rel:.1e})")
        assert rel < 2e-13 or np.max(np.abs(got - exact)) < 2e-10
