"""
The first-order tail (csrc/tail.hip) as mathematics, on the CPU: the backward recursion for the vector g --
what a deviation of the filter's MEANS at frame t does to the log-likelihood of all later frames -- against finite differences of a
plain NumPy Kalman filter; and the same for the matrix G of a deviation of the COVARIANCE, the lever DESIGN section 4 names as the one
that is left (derived there, not built on the device: this test pins the derivation for whoever builds it).
"""
import numpy as np
import pytest

import bild_amd


def _filter_tail(B, Sig, w, s2, x, C, M, t0):
    """ sum of the log-likelihoods of frames t0 .. T-1 of a switch-free filter whose state AFTER frame t0 - 1 is (C, M); NaN rows of x are
        missing frames (reference bild/src/MSRouse_logL_py.py:95-118: predict, masked update) """
    L = 0.0
    C, M = C.copy(), M.copy()
    for t in range(t0, len(x)):
        C = B @ C @ B.T + Sig
        M = B @ M
        if np.isnan(x[t, 0]):
            continue
        S = s2 + w @ C @ w
        K = C @ w / S
        e = x[t] - w @ M
        L += -0.5 * np.sum(np.log(2 * np.pi * S) + e * e / S)
        M = M + np.outer(K, e)
        C = C - np.outer(K, w @ C)
    return L


def _backward(B, Sig, w, s2, x, C0, M0):
    """ the table's filter forward (states after every frame), then g_t (N x d) and G_t (N x N) backward:
            g_{t-1} = B^T [ (e_t / S_t) w + (I - w K_t^T) g_t ]                                  (missing frame: B^T g_t)
            G_{t-1} = B^T [ a_t w w^T + A_t^T G_t A_t + sym(u_t w^T) ] B,   A_t = I - K_t w^T,
                      a_t = -1/2 (d / S_t - sum_m e_tm^2 / S_t^2),   u_t = A_t^T g_t (e_t / S_t)   (missing frame: B^T G_t B) """
    T, d = x.shape
    N = len(w)
    Cs, Ms = [C0.copy()], [M0.copy()]
    per_frame = [None]
    C, M = C0.copy(), M0.copy()
    for t in range(1, T):
        C = B @ C @ B.T + Sig
        M = B @ M
        if np.isnan(x[t, 0]):
            per_frame.append(None)
        else:
            S = s2 + w @ C @ w
            K = C @ w / S
            e = x[t] - w @ M
            per_frame.append((S, K, e))
            M = M + np.outer(K, e)
            C = C - np.outer(K, w @ C)
        Cs.append(C.copy()); Ms.append(M.copy())
    g = [None] * T
    G = [None] * T
    g[T - 1] = np.zeros((N, d)); G[T - 1] = np.zeros((N, N))
    for t in range(T - 1, 0, -1):
        if per_frame[t] is None:
            g[t - 1] = B.T @ g[t]
            G[t - 1] = B.T @ G[t] @ B
            continue
        S, K, e = per_frame[t]
        A = np.eye(N) - np.outer(K, w)
        g[t - 1] = B.T @ (np.outer(w, e / S) + A.T @ g[t])
        a = -0.5 * (d / S - np.sum(e * e) / S ** 2)
        u = A.T @ g[t] @ (e / S)
        Gm = a * np.outer(w, w) + A.T @ G[t] @ A + 0.5 * (np.outer(u, w) + np.outer(w, u))
        G[t - 1] = B.T @ Gm @ B
    return Cs, Ms, g, G


@pytest.mark.parametrize('N,T,missing', [(8, 60, ()), (12, 80, (5, 6, 7, 30, 79)), (6, 40, (39,))])
def test_tail_vectors_against_finite_differences(N, T, missing):
    rng = np.random.default_rng(N + T)
    model = bild_amd.MultiStateRouse(N, 1., 5., d=3, localization_error=0.1)
    A = model.arrays()
    B, Sig, C0, M0 = A['B'][1], A['Sig'][1], A['C0'][1], A['M0'][1]
    w = np.asarray(model.measurement, dtype=float)
    s2 = 0.1 ** 2
    x = np.array(model.trajectory_from_loopingprofile(bild_amd.Loopingprofile(np.ones(T, dtype=int)), rng=rng)[:], dtype=float)
    x[list(missing)] = np.nan
    Cs, Ms, g, G = _backward(B, Sig, w, s2, x, C0, M0)
    for t0 in (1, T // 3, T - 2):
        base = _filter_tail(B, Sig, w, s2, x, Cs[t0 - 1], Ms[t0 - 1], t0)
        # means: the first-order term is exact up to the quadratic one -- the error falls by 100 when the deviation falls by 10
        dM = rng.standard_normal(M0.shape)
        errs = []
        for eps in (1e-3, 1e-4):
            got = _filter_tail(B, Sig, w, s2, x, Cs[t0 - 1], Ms[t0 - 1] + eps * dM, t0) - base
            errs.append(abs(got - eps * np.sum(g[t0 - 1] * dM)))
        assert errs[0] < 1e-4 * max(1.0, abs(base)) and (errs[1] < 0.02 * errs[0] + 1e-12), (t0, errs)
        # covariance: a symmetric deviation inside the subspace the filter lives in
        R = rng.standard_normal((N, N))
        dC = Cs[t0 - 1] @ (R + R.T) @ Cs[t0 - 1]
        dC /= np.max(np.abs(dC))
        errs = []
        for eps in (1e-4, 1e-5):
            got = _filter_tail(B, Sig, w, s2, x, Cs[t0 - 1] + eps * dC, Ms[t0 - 1], t0) - base
            errs.append(abs(got - eps * np.sum(G[t0 - 1] * dC)))
        assert errs[1] < 0.02 * errs[0] + 1e-11 * max(1.0, abs(base)), (t0, errs)
        # ... and both at once: the cross term is second order too
        eps = 1e-5
        got = _filter_tail(B, Sig, w, s2, x, Cs[t0 - 1] + eps * dC, Ms[t0 - 1] + eps * dM, t0) - base
        first = eps * (np.sum(G[t0 - 1] * dC) + np.sum(g[t0 - 1] * dM))
        assert abs(got - first) < 1e-6 * max(1.0, abs(first) / eps), (t0, got, first)
