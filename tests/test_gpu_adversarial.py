"""
Data the model did not produce.  The convergence jumps compare a candidate's filter state with the
table's to 2^-43 RELATIVE to the column's scale, and for the mean columns that scale includes the data scale (largest
|coordinate|): the bound on the log-likelihood error per jump in DESIGN.md assumes |innovation| / S of order one, which
holds for data drawn from the model -- here it does not: a constant offset of 10^3 end-to-end distances, outlier frames,
a localization error of 10^-3, very soft and very stiff chains, long gaps.  Every case: tables + jumps (the default)
against the frame-by-frame run of the same kernel (jump=False: bit-identical to no tables at all) and against the CPU
oracle.  Bars: 1e-8 absolute, or -- where |logL| is so large (1e5 ... 5e9 here) that double precision itself is coarser --
64 ulp of |logL| between the two runs of the kernel and 2e-13 relative against the oracle (measured: <= 22 ulp = 5e-15, <= 5e-14).
"""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

CASES = {
    'offset_1e3':      dict(offset=1e3),
    'outliers_1pct':   dict(outliers=0.01),
    'sigma_1e-3':      dict(err=1e-3),
    'soft_chain':      dict(k=0.05),
    'stiff_chain':     dict(k=50.0),
    'long_gaps':       dict(gaps=True),
    'offset_outliers_gaps': dict(offset=300.0, outliers=0.01, gaps=True),
}


@pytest.mark.parametrize('case', sorted(CASES))
def test_jumps_on_data_the_model_did_not_produce(built_lib, case):
    import bild_amd
    from bild_amd import _lib
    from oracle import oracle
    c = CASES[case]
    rng = np.random.default_rng(sorted(CASES).index(case) + 40)
    T, n = 700, 3000
    err = c.get('err', 0.1)
    model = bild_amd.MultiStateRouse(20, 1.0, c.get('k', 5.0), d=3, localization_error=err)
    truth = H.random_profile(rng, T, 2, 120)
    miss = None
    if c.get('gaps'):
        mask = np.zeros(T, dtype=bool)
        for start in (60, 250, 470):
            mask[start:start + int(rng.integers(60, 120))] = True      # gaps longer than any transient
        miss = np.nonzero(mask)[0]
    traj = model.trajectory_from_loopingprofile(truth, missing_frames=miss, rng=rng)
    data = traj[:]
    scale = np.nanstd(data)
    if c.get('offset'):
        data += c['offset'] * scale                                    # the same offset in every dimension, every frame
    if c.get('outliers'):
        hit = rng.random(T) < c['outliers']
        data[hit] += 50.0 * scale * rng.standard_normal((int(hit.sum()), 3))
    model.invalidate()
    ss, thetas = H.candidate_profiles(rng, n, 5, 2)
    h, ts = model.handle(), model.trajset(traj)
    exact = _lib.logl_st(h, ts, ss, thetas, jump=False)               # every frame behind the first switch is run
    fast = _lib.logl_st(h, ts, ss, thetas)                            # tables + convergence jumps
    assert np.all(np.isfinite(exact)) and np.all(np.isfinite(fast))
    pick = rng.choice(n, 40, replace=False)
    want = oracle.logl_batch(model.arrays(), model.measurement, model.localization_error, data, H.expand(ss[pick], thetas[pick], T))
    ulp = np.spacing(np.abs(exact))
    dev = np.abs(fast - exact)
    bar = np.maximum(1e-8, 64 * ulp)
    worst = int(np.argmax(dev / bar))
    odev_fast, odev_exact = np.abs(fast[pick] - want), np.abs(exact[pick] - want)
    obar = np.maximum(1e-8, 2e-13 * np.abs(want))
    print(f"{case}: |logL| up to {np.max(np.abs(exact)):.3e} (ulp {ulp.max():.1e}); max |jumps - frame by frame| = {dev.max():.2e} "
          f"= {np.max(dev / ulp):.1f} ulp; against the oracle: jumps {odev_fast.max():.2e}, frame by frame {odev_exact.max():.2e} "
          f"(relative {np.max(odev_fast / np.abs(want)):.1e})")
    # (1) the jumps: 1e-8, or 64 ulp of |logL| where the log-likelihood is so large that its own rounding is coarser than that
    # (the tables hold RUNNING sums of the log-likelihood: every difference of two of them carries an ulp of the total)
    assert np.all(dev <= bar), (case, float(dev[worst]), float(bar[worst]))
    # (2) the oracle: 1e-8, or 2e-13 relative -- the modal reduction of the model (eigenbases to ~1e-14) shows as a RELATIVE
    # deviation of a few 1e-14, with and without the tables alike
    assert np.all(odev_fast <= obar), (case, float(odev_fast.max()))
    assert np.all(odev_exact <= obar), (case, float(odev_exact.max()))
