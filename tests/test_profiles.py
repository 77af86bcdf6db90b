"""
CPU tests of the host-side profile logic: Loopingprofile (pins of reference
tests/test_bild.py:51-121) and the (s, theta) -> switch-index encoding (pins of reference
tests/test_amis.py:199-202 and the slice semantics of bild/amis.py:685-693).
"""
import numpy as np

import helpers as H
from bild_amd.profiles import (Loopingprofile, state_probabilities, switch_indices, segments_from_st,
                               segments_from_states, states_from_segments)
from bild_amd.amis import FixedkSampler


def test_loopingprofile_basics():
    p = Loopingprofile([0, 0, 0, 1, 1, 0])
    assert len(p) == 6 and p[3] == 1 and p.count_switches() == 2
    q = p.copy()
    assert q == p
    q[0] = 1
    assert not (q == p) and p[0] == 0
    assert p.intervals() == [(None, 3, 0), (3, 5, 1), (5, None, 0)]
    t, y = p.plottable()
    assert np.array_equal(t, [-1, 2, 2, 4, 4, 5]) and np.array_equal(y, [0, 0, 1, 1, 0, 0])
    try:
        p[0] = 1.5
        raise RuntimeError("float assignment must be rejected")
    except AssertionError:
        pass
    assert not (p == Loopingprofile([0, 0]))


def test_state_probabilities():
    profs = [Loopingprofile([0, 0, 1]), Loopingprofile([0, 1, 1])]
    pr = state_probabilities(profs)
    assert np.allclose(pr, [[1, .5, 0], [0, .5, 1]])
    assert state_probabilities(profs, nStates=3).shape == (3, 3)


def _FakeModel(T=None, n=3):
    """ likelihood double: the sampler's constructor may evaluate profiles exhaustively """
    from amis_cases import TableModel
    return TableModel(np.zeros((n, 400)))


def test_st2profile_reference_pin():
    # reference tests/test_amis.py:199-202: st2profile([.25,.5,.25],[0,1,0]) on T=6 -> [0,0,1,1,0,0]
    sampler = FixedkSampler(np.zeros((6, 1)), _FakeModel(), k=2, max_fcomplete=0)
    prof = sampler.st2profile(np.array([0.25, 0.5, 0.25]), np.array([0, 1, 0]))
    assert np.array_equal(prof[:], [0, 0, 1, 1, 0, 0])
    assert isinstance(prof, Loopingprofile) and prof.state.dtype.kind == 'i'


def test_segments_match_st2profile_semantics():
    rng = np.random.default_rng(0)
    for T in (2, 5, 37, 200):
        for k in (0, 1, 3, 7):
            ss, thetas = H.candidate_profiles(rng, 50, k, 3)
            # force degenerate cases: zero-length intervals and a cumsum that reaches 1.0 early
            ss[0, :] = 0
            ss[0, 0] = 1.0
            if k >= 2:
                ss[1, 1] = 0.0
                ss[1] /= ss[1].sum()
            sampler = FixedkSampler(np.zeros((T, 1)), _FakeModel(), k=k, max_fcomplete=(1000 if k == 0 else 3))
            seg_start, seg_state = segments_from_st(ss, thetas, T)
            assert seg_start.dtype == np.int32 and np.all(seg_start[:, 0] == 0)
            assert np.all(np.diff(seg_start, axis=1) >= 0) and np.all(seg_start[:, 1:] >= 1)
            exp = states_from_segments(seg_start, seg_state, T)
            for r in range(len(ss)):
                assert np.array_equal(sampler.st2profile(ss[r], thetas[r])[:], exp[r])
            # run-length encoding of the expanded profiles gives the same profiles back
            a, b = segments_from_states(exp)
            assert np.array_equal(states_from_segments(a, b, T), exp)


def test_switch_indices_formula():
    ss = np.array([[0.25, 0.5, 0.25], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
    assert np.array_equal(switch_indices(ss, 6), [[2, 4], [1, 1], [6, 6]])
    assert switch_indices(np.ones((4, 1)), 10).shape == (4, 0)


def test_loopingprofile_reference_fixture():
    """ reference tests/test_bild.py:51-121 on its own fixture """
    assert np.array_equal(Loopingprofile().state, np.array([]))
    assert np.array_equal(Loopingprofile([1, 2, 3]).state, [1, 2, 3])
    p = Loopingprofile([0, 0, 0, 1, 1, 0, 3, 3])
    q = p.copy()
    assert np.array_equal(q.state, p.state)
    q[2] = 5
    assert p[2] == 0                                                    # :61-65
    assert len(p) == 8 and p[3] == 1 and np.array_equal(p[2:4], [0, 1])  # :67-73
    p[2] = 3
    assert p[2] == 3
    try:
        p[5] = 3.74
        raise RuntimeError("float assignment must be rejected")         # :78-79
    except AssertionError:
        pass
    assert p == Loopingprofile([0, 0, 3, 1, 1, 0, 3, 3]) and p != Loopingprofile([1, 0, 3])
    p = Loopingprofile([0, 0, 0, 1, 1, 0, 3, 3])
    assert p.count_switches() == 3                                      # :85-90
    p[5] = 1
    assert p.count_switches() == 2
    p[4] = 2
    assert p.count_switches() == 4
    p = Loopingprofile([0, 0, 0, 1, 1, 0, 3, 3])
    assert p.intervals() == [(None, 3, 0), (3, 5, 1), (5, 6, 0), (6, None, 3)]   # :92-102
    assert Loopingprofile([1, 1, 1, 1]).intervals() == [(None, None, 1)]
    t, y = p.plottable()                                                # :104-107
    assert np.array_equal(t, [-1, 2, 2, 4, 4, 5, 5, 7]) and np.array_equal(y, [0, 0, 1, 1, 0, 0, 3, 3])
    profs = [Loopingprofile([0, 1, 0, 1, 0]), Loopingprofile([1, 1, 1, 1, 1])]   # :109-121
    assert np.array_equal(state_probabilities(profs), [[0.5, 0, 0.5, 0, 0.5], [0.5, 1, 0.5, 1, 0.5]])
    assert np.array_equal(state_probabilities(profs, nStates=3),
                          [[0.5, 0, 0.5, 0, 0.5], [0.5, 1, 0.5, 1, 0.5], [0, 0, 0, 0, 0]])


def _st_golden():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'st2profile.npz'))
    for i in range(int(z['n_cases'])):
        yield (z[f'ss_{i}'], z[f'thetas_{i}'], int(z[f'T_{i}']), int(z[f'S_{i}']), z[f'states_{i}'].astype(np.int64))


def test_st_encoding_against_the_reference_vectors(built_lib):
    """
    Profiles returned by the REFERENCE's FixedkSampler.st2profile (tests/golden/make_st_golden.py) against
    (i) the NumPy statement of the encoding, (ii) the native conversion behind bild_logl_st, (iii) this package's
    st2profile.  Bit-exact integer work.
    """
    from bild_amd import _lib
    n_total = 0
    for ss, thetas, T, S, want in _st_golden():
        a, b = segments_from_st(ss, thetas, T)
        assert np.array_equal(states_from_segments(a, b, T), want)
        na, nb = _lib.segments_from_st(ss, thetas, T, S)
        assert np.array_equal(na, a) and np.array_equal(nb, b)
        sampler = FixedkSampler.__new__(FixedkSampler)
        sampler.traj = np.zeros((T, 1))
        for r in range(0, len(ss), 5):
            assert np.array_equal(sampler.st2profile(ss[r], thetas[r])[:], want[r])
        n_total += len(ss)
    assert n_total > 1000


def test_native_st_conversion_equals_numpy_on_random_batches(built_lib):
    """ sequential cumsum, one multiply, floor: the native loop and NumPy agree in every bit, also per-sample T """
    from bild_amd import _lib
    rng = np.random.default_rng(5)
    for k in (0, 1, 4, 20):
        n = 4000
        ss = rng.dirichlet(rng.choice([0.05, 1.0, 30.0]) * np.ones(k + 1), size=n)
        thetas = rng.integers(3, size=(n, k + 1))
        Ts = rng.integers(1, 3000, size=n).astype(np.int32)
        na, nb = _lib.segments_from_st(ss, thetas, Ts, 3)
        for T in np.unique(Ts)[::97]:
            sel = Ts == T
            a, b = segments_from_st(ss[sel], thetas[sel], int(T))
            assert np.array_equal(na[sel], a) and np.array_equal(nb[sel], b)
    # refused: states out of range, negative or non-finite interval lengths
    import pytest
    good_s, good_t = np.array([[0.5, 0.5]]), np.array([[0, 1]])
    for bad_s, bad_t in ((good_s, np.array([[0, 3]])), (np.array([[np.nan, 0.5]]), good_t), (np.array([[-0.5, 1.5]]), good_t)):
        with pytest.raises(_lib.BildAmdError):
            _lib.segments_from_st(bad_s, bad_t, 100, 3)
