#!/usr/bin/env python3
"""
Generates the golden vectors under tests/golden/ from the REFERENCE's own kernels.

Runs only in the build container (needs /root/reference):
  * reference NumPy kernel  bild/src/MSRouse_logL_py.py  (loaded by path)
  * reference Cython kernel bild/src/MSRouse_logL.pyx    (compiled unmodified by oracle/build_ref.py)
Both are fed duck-typed model / profile / trajectory objects built from this package's Rouse
matrix builder (the `rouse` package the reference would use is not installed).

Each fixture is DATA: the array inputs the kernel consumes (B, G, Sig, M0, C0, w, localization
error, trajectory, expanded profiles) and the two reference outputs.  No reference source text
is stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.dont_write_bytecode = True

import helpers as H  # noqa: E402
from bild_amd.trajectory import Trajectory  # noqa: E402
from oracle import oracle  # noqa: E402

ref_numpy = oracle.load_reference_numpy()
ref_cython = oracle.load_reference_cython()
assert ref_numpy is not None and ref_cython is not None, "reference kernels unavailable (run in the build container)"


def evaluate(model, traj, states_batch):
    out_np, out_cy = [], []
    all_missing = not np.any(~np.any(np.isnan(traj[:]), axis=1))
    for st in states_batch:
        prof = H.ProfileView(st)
        out_np.append(ref_numpy(model, prof, traj))
        # the Cython kernel reads valid_times[0] out of bounds on an all-missing trajectory
        # (pyx:186): only the NumPy form defines that case (-> 0.0)
        out_cy.append(np.nan if all_missing else ref_cython(model, prof, traj))
    return np.array(out_np), np.array(out_cy)


def save(name, model, traj, states_batch, note, extra=None):
    states_batch = np.atleast_2d(np.asarray(states_batch, dtype=np.int64))
    out_np, out_cy = evaluate(model, traj, states_batch)
    a = model.arrays()
    payload = dict(B=a['B'], G=a['G'], Sig=a['Sig'], M0=a['M0'], C0=a['C0'], w=model.measurement,
                   localization_error=np.asarray(model._get_noise(traj), dtype=float),
                   x=np.asarray(traj[:], dtype=float), states=states_batch.astype(np.int16),
                   logL_ref_numpy=out_np, logL_ref_cython=out_cy, note=np.array(note))
    if extra:
        payload.update(extra)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **payload)
    print(f"{name:28s} n={len(out_np):3d} T={states_batch.shape[1]:5d} "
          f"max|cy-np|={np.nanmax(np.abs(out_cy - out_np)) if np.any(~np.isnan(out_cy)) else float('nan'):.2e} "
          f"{os.path.getsize(path) / 1024:.1f} KB")


def main():
    # 1. the reference's own unit-test fixture (tests/test_bild.py:125-138):
    #    Trajectory([1, 2, nan, 4], localization_error=[0.5]), profile [1, 1, 0, 0],
    #    MultiStateRouse(20, 1, 5, d=1)
    model = H.DuckModel(N=20, D=1, k=5, d=1)
    traj = Trajectory([1, 2, np.nan, 4], localization_error=[0.5])
    save('ref_unittest_4frames', model, traj, [[1, 1, 0, 0], [0, 0, 0, 0], [1, 0, 1, 0]],
         "reference tests/test_bild.py:125-138 fixture; the test pins -100 < logL < 0 and Cython == NumPy")

    # 2. 2-state, d = 3, one localization error (d* = 1), T = 200, AMIS-style candidates
    rng = np.random.default_rng(20241008)
    model = H.DuckModel(N=20, D=1, k=5, d=3, localization_error=0.1)
    truth = H.random_profile(rng, 200, 2, 40)
    traj = H.synth_trajectory(model, truth, 0.1, rng)
    batch = [truth, np.zeros(200, int), np.ones(200, int)]
    for k in (1, 2, 4, 8):
        ss, th = H.candidate_profiles(rng, 3, k, 2)
        batch += list(H.expand(ss, th, 200))
    # adjacent equal switch indices (a state of zero length) and a switch at the very end
    ss = np.array([[0.3, 0.0, 0.7], [0.5, 0.5 - 1e-12, 1e-12]])
    th = np.array([[0, 1, 0], [1, 0, 1]])
    batch += list(H.expand(ss, th, 200))
    save('s2_d3_T200', model, traj, batch, "2-state N=20 d=3 d*=1 T=200, k in {0,1,2,4,8}, empty segments")

    # 3. d* = 2 (z error differs), missing frames incl. frame 0 and trailing frames
    rng = np.random.default_rng(7)
    err = np.array([0.1, 0.1, 0.25])
    model = H.DuckModel(N=20, D=1, k=5, d=3, localization_error=err)
    truth = H.random_profile(rng, 150, 2, 30)
    missing = np.concatenate([[0, 1], np.arange(60, 85), H.missing_mask(rng, 150, 'iid'), [147, 148, 149]])
    traj = H.synth_trajectory(model, truth, err, rng, missing=missing)
    ss, th = H.candidate_profiles(rng, 6, 3, 2)
    save('s2_dstar2_missing_T150', model, traj, [truth] + list(H.expand(ss, th, 150)),
         "d*=2 (errors .1,.1,.25), frame 0 missing, 25-frame gap, trailing frames missing")

    # 4. d* = 3, unsorted errors (np.unique sorts them: pyx:145)
    rng = np.random.default_rng(8)
    err = np.array([0.3, 0.05, 0.2])
    model = H.DuckModel(N=12, D=0.5, k=2, d=3, localization_error=err)
    truth = H.random_profile(rng, 80, 2, 20)
    traj = H.synth_trajectory(model, truth, err, rng, missing=H.missing_mask(rng, 80, 'iid'))
    ss, th = H.candidate_profiles(rng, 4, 2, 2)
    save('s2_dstar3_N12_T80', model, traj, [truth] + list(H.expand(ss, th, 80)), "d*=3 unsorted errors, N=12")

    # 5. 3-state model (free / end-to-end loop / 0-10 loop), bursty missing frames, T = 300
    rng = np.random.default_rng(9)
    model = H.DuckModel(N=20, D=1, k=5, d=3, loops=H.LOOPS[3], localization_error=0.1)
    truth = H.random_profile(rng, 300, 3, 50)
    traj = H.synth_trajectory(model, truth, 0.1, rng, missing=H.missing_mask(rng, 300, 'bursty'))
    ss, th = H.candidate_profiles(rng, 8, 5, 3)
    save('s3_bursty_T300', model, traj, [truth] + list(H.expand(ss, th, 300)), "3-state, bursty missing frames")

    # 6. all frames missing: NumPy form returns 0.0 (MSRouse_logL_py.py:90-94); Cython is undefined there
    model = H.DuckModel(N=8, D=1, k=5, d=2, localization_error=0.1)
    traj = Trajectory(np.full((10, 2), np.nan), localization_error=[0.1, 0.1])
    save('all_missing_T10', model, traj, [np.zeros(10, int), np.arange(10) % 2], "all frames missing -> 0.0")

    # 7. measurement vector with sum(w) != 0 and only some coordinates missing in a frame
    rng = np.random.default_rng(10)
    w = np.zeros(10)
    w[2], w[7] = -1.0, 0.5
    model = H.DuckModel(N=10, D=1, k=3, d=2, loops=(None, (2, 7)), localization_error=0.05, measurement=w)
    truth = H.random_profile(rng, 60, 2, 15)
    traj = H.synth_trajectory(model, truth, 0.05, rng)
    traj.data[5, 0] = np.nan     # one coordinate missing -> the whole frame is missing (pyx:178)
    traj.data[17, 1] = np.nan
    ss, th = H.candidate_profiles(rng, 4, 2, 2)
    save('asym_w_partial_nan_T60', model, traj, [truth] + list(H.expand(ss, th, 60)),
         "sum(w) != 0, interior loop, partially-NaN frames, d=2")

    # 8. headline shape, T = 1000 (two samples only: file size)
    rng = np.random.default_rng(11)
    model = H.DuckModel(N=20, D=1, k=5, d=3, localization_error=0.1)
    truth = H.random_profile(rng, 1000, 2, 200)
    traj = H.synth_trajectory(model, truth, 0.1, rng)
    ss, th = H.candidate_profiles(rng, 3, 4, 2)
    save('s2_d3_T1000', model, traj, [truth] + list(H.expand(ss, th, 1000)), "headline shape N=20 d=3 d*=1 T=1000 2-state")

    # 9. external force on the chain: G != 0 and a non-zero steady-state mean M0 (never produced by the
    #    reference's own constructor, which builds rouse.Model with F = 0, but the kernels accept it)
    rng = np.random.default_rng(12)
    model = H.DuckModel(N=20, D=1, k=5, d=3, localization_error=[0.1, 0.1, 0.2])
    for mi, mod in enumerate(model.models):
        mod.F[0, :] = [0.5, -0.25, 0.1 * (mi + 1)]
        mod.F[-1, :] = [-0.5, 0.25, -0.1 * (mi + 1)]
        mod.F[7, 1] = 0.3
        mod.update_dynamics()
    truth = H.random_profile(rng, 120, 2, 30)
    traj = H.synth_trajectory(model, truth, [0.1, 0.1, 0.2], rng, missing=H.missing_mask(rng, 120, 'iid'))
    ss, th = H.candidate_profiles(rng, 5, 3, 2)
    save('force_G_T120', model, traj, [truth] + list(H.expand(ss, th, 120)), "non-zero G and M0 (external force), d*=2")


if __name__ == '__main__':
    main()
