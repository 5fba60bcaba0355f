"""
Synthetic data recipes for the collapsed-Gibbs path.

Every recipe is a pure function of its integer seed (legacy ``np.random``
MT19937 streams, which are stable across numpy versions), so golden fixtures
store only the recipe name + arguments and a checksum of ``X``.

Reference shapes these reproduce:
  * ``gendata_1d``      -- pybgmm/utils/gendata.py:3-25 (config C1)
  * ``demo_mixture``    -- examples/crpmm_2d_demo.py:42-48 and
                           pybgmm/tests/test_igmm.py:32-38 (column-major draw order)
  * ``synth_mixture``   -- SURVEY.md section 8(d): the bench workload
  * ``demo_prior``      -- examples/crpmm_2d_demo.py:51-55
"""
import hashlib

import numpy as np


def gendata_1d(N, W=(0.35, 0.4, 0.25), MU=(0., 2., 5.), SIGMA=(0.5, 0.5, 1.), seed=12345):
    """Three-component 1-D mixture; returns ``(MU, X[N,1], y[N])``."""
    rs = np.random.RandomState(seed)
    W = np.asarray(W, dtype=float)
    assert np.sum(W) == 1, "weight vector should sum to 1"
    MU = np.asarray(MU, dtype=float)
    SIGMA = np.asarray(SIGMA, dtype=float)
    y = rs.choice(MU.size, size=N, p=W)
    X = rs.normal(MU[y], SIGMA[y], size=N)
    return MU, X.reshape((N, 1)), y


def demo_mixture(N, D, K_true, mu_scale=4.0, covar_scale=0.7, rs=None):
    """
    The demos' generator: ``z_true = randint``, ``mu = randn(D, K)``,
    ``X = (mu[:, z] + randn(D, N) * s).T``.  ``rs`` is a ``RandomState`` (or the
    ``np.random`` module itself, to consume the caller's global stream exactly
    like the reference scripts do).
    """
    rs = np.random if rs is None else rs
    z_true = rs.randint(0, K_true, N)
    mu = rs.randn(D, K_true) * mu_scale
    X = mu[:, z_true] + rs.randn(D, N) * covar_scale
    return np.ascontiguousarray(X.T), z_true


def synth_mixture(N, D, K_true, seed, mu_scale=4.0, covar_scale=0.7):
    """
    Bench workload: isotropic mixture, balanced shuffled labels.
    Returns ``(X[N,D] float64 C-contiguous, z_true[N] int64)``.
    """
    rs = np.random.RandomState(seed)
    mu = rs.randn(K_true, D) * mu_scale
    z_true = np.arange(N) % K_true
    rs.shuffle(z_true)
    X = mu[z_true]
    # chunked noise so that N=2e6, D=128 does not need a second full-size temp
    step = max(1, (1 << 24) // max(D, 1))
    for lo in range(0, N, step):
        hi = min(N, lo + step)
        X[lo:hi] += rs.randn(hi - lo, D) * covar_scale
    return np.ascontiguousarray(X), z_true.astype(np.int64)


def demo_prior_params(D, mu_scale=4.0, covar_scale=0.7, v_0=None):
    """``(m_0, k_0, v_0, S_0)`` exactly as the demo scripts build them."""
    m_0 = np.zeros(D)
    k_0 = covar_scale ** 2 / mu_scale ** 2
    v_0 = D + 3 if v_0 is None else v_0
    S_0 = covar_scale ** 2 * v_0 * np.eye(D)
    return m_0, k_0, v_0, S_0


def array_digest(a):
    """sha256 over the raw little-endian bytes of a C-contiguous array."""
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()
