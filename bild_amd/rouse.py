"""
Rouse-chain dynamics: the per-state matrices the Kalman-filter likelihood consumes.

The reference obtains these from the un-vendored third-party package ``rouse``
(``rouse.Model(N, D, k, d, add_bonds=loop)``, reference bild/models.py:246) and reads
``_dynamics['B'|'G'|'Sig']`` and ``steady_state()`` from it
(bild/src/MSRouse_logL.pyx:152-160). ``rouse`` is not available to this build, so this
module restates the standard discrete-time Rouse / Ornstein-Uhlenbeck physics from first
principles; it exposes the same duck-typed surface the reference kernels touch
(``check_dynamics``, ``_dynamics``, ``steady_state``, ``propagate_M``, ``propagate_C``,
``conf_ss``, ``evolve``), so a `Model` here can be handed to the reference's
``MSRouse_logL`` unchanged.  Parity with an installed ``rouse`` is *unpinned* (SURVEY.md
section 8c, "Level M"); the kernel-level parity contract is on identical array inputs.

Physics (one spatial dimension; all d dimensions are independent and identical):

    dx = -k A x dt + F dt + sqrt(2 D) dW

with ``A`` the connectivity (graph-Laplacian) matrix of the chain plus extra bonds.  With
the symmetric eigendecomposition ``k A = V diag(a) V^T`` the exact one-frame (dt = 1)
propagator is

    B   = V diag(exp(-a))                V^T
    Sig = V diag(D (1 - exp(-2a)) / a)   V^T      (a -> 0:  2 D)
    G   = V diag((1 - exp(-a)) / a)      V^T F    (a -> 0:  F)

and the steady state is ``M0 = (kA)^+ F``, ``C0 = D (kA)^+`` on the internal modes.  The
free centre-of-mass mode (a = 0) has no steady state; **convention of this build**: it is
pinned (variance 0, mean 0), i.e. the Moore-Penrose pseudo-inverse is used.  For any
measurement vector with ``sum(w) == 0`` (the reference default, end-to-end) the likelihood
does not depend on that choice, because the uniform vector is a common null vector of
every state's Laplacian.
"""
import numpy as np

# relative threshold under which an eigenvalue of k*A counts as a zero (free) mode
_ZERO_MODE_RTOL = 1e-12


class Model:
    """
    One Rouse model (one looping state).

    Parameters mirror ``rouse.Model(N, D, k, d, add_bonds=...)`` as called at reference
    bild/models.py:246.

    Parameters
    ----------
    N : int
        number of monomers
    D, k : float
        monomer diffusion constant and backbone spring constant
    d : int
        spatial dimension
    add_bonds : list of (i, j[, rel_strength]) or None
        extra harmonic bonds of strength ``rel_strength * k`` between monomers ``i`` and
        ``j`` (negative indices count from the end, as in reference bild/models.py:223).
    """

    def __init__(self, N, D=1., k=1., d=3, add_bonds=None):
        self.N = int(N)
        self.D = float(D)
        self.k = float(k)
        self.d = int(d)

        A = 2. * np.eye(self.N) - np.eye(self.N, k=1) - np.eye(self.N, k=-1)
        A[0, 0] = A[-1, -1] = 1.
        if self.N == 1:
            A[:] = 0.
        self.A = A
        self.F = np.zeros((self.N, self.d))

        if add_bonds is not None:
            for bond in add_bonds:
                self.add_bond(*bond)

        self._dynamics = {'needs_updating': True}

    def add_bond(self, i, j, rel_strength=1.):
        i = int(i) % self.N
        j = int(j) % self.N
        if i != j:
            self.A[i, i] += rel_strength
            self.A[j, j] += rel_strength
            self.A[i, j] -= rel_strength
            self.A[j, i] -= rel_strength
        self._dynamics = {'needs_updating': True}

    # ------------------------------------------------------------------ dynamics
    def update_dynamics(self):
        a, V = np.linalg.eigh(self.k * self.A)
        scale = max(np.max(np.abs(a)), 1e-300)
        if np.any(a < -_ZERO_MODE_RTOL * scale * self.N):
            raise ValueError("connectivity matrix has a negative mode: dynamics are unstable")
        zero = np.abs(a) <= _ZERO_MODE_RTOL * scale * self.N
        a_safe = np.where(zero, 1., a)

        b = np.where(zero, 1., np.exp(-a_safe))
        g = np.where(zero, 1., -np.expm1(-a_safe) / a_safe)
        sig = np.where(zero, 2. * self.D, -self.D * np.expm1(-2. * a_safe) / a_safe)
        cinf = np.where(zero, 0., self.D / a_safe)
        minf = np.where(zero, 0., 1. / a_safe)

        def sym(diag):
            X = (V * diag) @ V.T
            return 0.5 * (X + X.T)

        self._dynamics = {
            'needs_updating': False,
            'N': self.N, 'D': self.D, 'k': self.k, 'd': self.d,
            'B': np.ascontiguousarray(sym(b)),
            'G': np.ascontiguousarray(sym(g) @ self.F),
            'Sig': np.ascontiguousarray(sym(sig)),
            'M0': np.ascontiguousarray(sym(minf) @ self.F),
            'C0': np.ascontiguousarray(sym(cinf)),
            # square root of Sig / C0 for sampling
            'LSig': V * np.sqrt(np.maximum(sig, 0.)),
            'LC0': V * np.sqrt(np.maximum(cinf, 0.)),
        }

    def check_dynamics(self, run_if_necessary=True):
        if self._dynamics['needs_updating']:
            if not run_if_necessary:
                raise RuntimeError("Model changed since last call to update_dynamics()")
            self.update_dynamics()

    def steady_state(self):
        """ -> (M (N, d), C (N, N)) ; reference call sites pyx:160, models.py:366 """
        self.check_dynamics()
        return self._dynamics['M0'].copy(), self._dynamics['C0'].copy()

    # reference call sites: bild/src/MSRouse_logL_py.py:109-110
    def propagate_M(self, M, check_dynamics=True):
        if check_dynamics:
            self.check_dynamics()
        return self._dynamics['B'] @ M + self._dynamics['G']

    def propagate_C(self, C, check_dynamics=True):
        if check_dynamics:
            self.check_dynamics()
        B = self._dynamics['B']
        return B @ C @ B + self._dynamics['Sig']

    # reference call sites: bild/models.py:332,337 (generative model)
    def conf_ss(self, rng=None):
        self.check_dynamics()
        rng = np.random.default_rng() if rng is None else rng
        return self._dynamics['M0'] + self._dynamics['LC0'] @ rng.standard_normal((self.N, self.d))

    def evolve(self, conf, rng=None):
        self.check_dynamics()
        rng = np.random.default_rng() if rng is None else rng
        return (self._dynamics['B'] @ conf + self._dynamics['G']
                + self._dynamics['LSig'] @ rng.standard_normal((self.N, self.d)))


def stack_dynamics(models):
    """
    Stack the per-state arrays the likelihood path consumes
    (reference bild/src/MSRouse_logL.pyx:152-160).

    Returns
    -------
    dict with B (S,N,N), G (S,N,d), Sig (S,N,N), M0 (S,N,d), C0 (S,N,N), all C-contiguous f64
    """
    for m in models:
        m.check_dynamics()
    out = {}
    for key in ('B', 'G', 'Sig', 'M0', 'C0'):
        out[key] = np.ascontiguousarray(np.stack([m._dynamics[key] for m in models]), dtype=np.float64)
    return out
