"""
GPU tests of the library's experiment switches (csrc/config.h): a parity subset run once under each switch that changes
which kernels / tables serve a batch.  Round 3's NaN-row bug was an untested switch combination; this is the systematic
version of what was then done by hand (the whole suite under BILD_NO_SPLIT=1).

The switches are read once per process; `bild_config_reload` re-reads them, so one process covers them all.
"""
import os

import numpy as np
import pytest

import goldens
import helpers as H

pytestmark = pytest.mark.gpu

TOL = 1e-8
SWITCHES = ['BILD_NO_SPLIT', 'BILD_NO_STATES', 'BILD_NO_LISTED_GEOMETRY', 'BILD_NO_PAIRS', 'BILD_NO_TRANSIENTS', 'BILD_NO_JUMP',
            'BILD_NO_PREFIX', 'BILD_NO_WALK_PLAN', 'BILD_NO_FUSED_LAUNCH', 'BILD_NO_SPLIT+BILD_NO_STATES',
            'BILD_NO_LISTED_GEOMETRY+BILD_NO_STATES', 'BILD_NO_TAIL', 'BILD_NO_TAIL+BILD_NO_STATES', 'BILD_TAIL_TOL_BITS=24',
            'BILD_TAIL_MARGIN=0', 'BILD_STATES_STRIDE=1', 'BILD_STATES_MAX_GAP=16', 'BILD_STATES_MAX_BYTES=1000000',
            'BILD_TABLE_CACHE_BYTES=0']


@pytest.fixture
def switch(request, built_lib):
    from bild_amd import _lib
    settings = [(item.split('=') + ['1'])[:2] for item in request.param.split('+')]      # NAME or NAME=value
    for name, value in settings:
        os.environ[name] = value
    _lib.config_reload()
    assert all(f'{name}={value}' in _lib.config_string().split() for name, value in settings)
    yield request.param
    for name, _ in settings:
        del os.environ[name]
    _lib.config_reload()


def _subset(tag):
    """ goldens, sampler batches against the oracle (tables, masks, three states, d* = 2), refused rows, a 16-mode chain """
    import bild_amd
    from bild_amd import _lib
    from oracle import oracle
    # (i) the reference's own vectors
    for name in ('s2_d3_T200', 's2_dstar2_missing_T150', 's3_bursty_T300'):
        g = goldens.load(name)
        m = bild_amd.MultiStateRouse.from_arrays(g['B'], g['G'], g['Sig'], g['M0'], g['C0'], g['w'],
                                                 localization_error=g['localization_error'])
        got = m.logL_batch(g['states'], g['x'])
        for key in ('logL_ref_numpy', 'logL_ref_cython'):
            ok = ~np.isnan(g[key])
            assert np.max(np.abs(got[ok] - g[key][ok])) < TOL, (tag, name, key)
    # (ii) batches large enough for every table to be built and used, against the oracle
    for S, T, k, miss, err, N in ((2, 400, 4, 'none', 0.1, 20), (3, 300, 5, 'bursty', [0.1, 0.1, 0.3], 20), (2, 250, 8, 'iid', 0.1, 32)):
        rng = np.random.default_rng(7 * S + T + k)
        model = bild_amd.MultiStateRouse(N, 1, 5, d=3, looppositions=H.LOOPS[S], localization_error=err)
        traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, S, max(T // 5, 2)),
                                                    missing_frames=H.missing_mask(rng, T, miss), rng=rng)
        ss, thetas = H.candidate_profiles(rng, 4000, k, S)
        got = bild_amd.FixedkSampler(traj, model, k=k, N=len(ss), max_fcomplete=0).logL(ss, thetas)
        pick = rng.choice(len(ss), 96, replace=False)
        want = oracle.logl_batch(model.arrays(), model.measurement, model.localization_error, traj[:], H.expand(ss[pick], thetas[pick], T))
        assert np.max(np.abs(got[pick] - want)) < TOL, (tag, S, T, k)
        # a second evaluation of the same rows is bit-identical (tables built at the first one, never later)
        again = model.logL_st_batch(ss, thetas, traj)
        assert np.array_equal(got, again), tag
        # (iii) a row that is no point on the simplex: refused by the host entry, NaN (and reported) by the device entry
        bad = ss.copy()
        bad[5, 0] = -1e-9
        with pytest.raises(_lib.BildAmdError):
            model.logL_st_batch(bad, thetas, traj)


@pytest.mark.parametrize('switch', SWITCHES, indirect=True)
def test_parity_subset_under_switch(switch):
    _subset(switch)


def test_parity_subset_defaults(built_lib):
    from bild_amd import _lib
    assert _lib.config_string() == '' or 'BILD_' in _lib.config_string()
    _subset('defaults')


def test_negative_position_on_the_device_entry(built_lib):
    """ walk.hip refuses a negative cumulative position like the host conversion does (NaN + status), tables or not """
    import torch
    import bild_amd
    from bild_amd import _lib
    rng = np.random.default_rng(3)
    model = bild_amd.MultiStateRouse(20, 1, 5, d=3, localization_error=0.1)
    T = 300
    traj = model.trajectory_from_loopingprofile(H.random_profile(rng, T, 2, 60), rng=rng)
    ss, thetas = H.candidate_profiles(rng, 3000, 3, 2)
    ref = model.logL_st_batch(ss, thetas, traj)               # builds the tables
    ss[17, 0] = -1e-9                                           # cumulative position in (-1, 0)
    dev = torch.device('cuda', 0)
    d_ss = torch.from_numpy(ss).to(dev)
    d_th = torch.from_numpy(thetas.astype(np.uint8)).to(dev)
    d_out = torch.zeros(len(ss), dtype=torch.float64, device=dev)
    d_status = torch.zeros(2, dtype=torch.int32, device=dev)
    _lib.logl_st_device(model.handle(), model.trajset(traj), len(ss), 4, d_ss.data_ptr(), d_th.data_ptr(), d_out.data_ptr(),
                        stream=torch.cuda.current_stream().cuda_stream, d_status=d_status.data_ptr())
    torch.cuda.synchronize()
    out = d_out.cpu().numpy()
    assert np.isnan(out[17]) and d_status.cpu().tolist() == [1, 17]
    keep = np.ones(len(ss), dtype=bool)
    keep[17] = False
    assert np.array_equal(out[keep], ref[keep])
