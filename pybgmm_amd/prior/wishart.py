"""
Wishart / inverse-Wishart draws by the Bartlett decomposition -- the interface of reference
pybgmm/prior/wishart.py:16-32 (``wishrnd``, ``iwishrnd``), used by ``GaussianComponents.rand_k``.

The caller-visible random streams are consumed exactly as there: row r of the Bartlett factor takes
``np.random.normal(size=(r,))`` (rows r >= 1) and then ONE ``random.gammavariate(0.5 (v_0 - D + 1), 2.0)``
for its diagonal entry -- the same shape parameter on every row, and the factor is held in float32,
both as in the reference.  ``rng`` / ``nprng`` replace the process-global streams (one chain per GPU).
"""
import math
import random as _random

import numpy as np


def wishrnd(sigma, v_0, C=None, rng=None, nprng=None):
    """A sample from a Wishart distribution (C: a factor of sigma, Cholesky by default)."""
    rng = _random if rng is None else rng
    nprng = np.random if nprng is None else nprng
    if C is None:
        C = np.linalg.cholesky(sigma)
    D = sigma.shape[0]
    a = np.zeros((D, D), dtype=np.float32)
    for r in range(D):
        if r != 0:
            a[r, :r] = nprng.normal(size=(r,))
        a[r, r] = math.sqrt(rng.gammavariate(0.5 * (v_0 - D + 1), 2.0))
    return np.dot(np.dot(np.dot(C, a), a.T), C.T)


def iwishrnd(sigma, v_0, C=None, rng=None, nprng=None):
    """A sample from an inverse-Wishart distribution."""
    sample = wishrnd(sigma, v_0, C, rng=rng, nprng=nprng)
    return np.linalg.solve(sample, np.eye(sample.shape[0]))
