"""
Wishart / inverse-Wishart draws (Bartlett construction) behind ``GaussianComponents.rand_k``.

Interface of the reference's pybgmm/prior/wishart.py:16-32 (``wishrnd(sigma, v_0, C)``,
``iwishrnd(sigma, v_0, C)``); what is kept from there is only what a caller can observe:

* the two caller-visible generators are consumed in the same amounts and order -- the strictly lower
  triangle of the Bartlett factor is D (D - 1) / 2 normals of ``np.random`` in row-major order, its
  diagonal D draws of ``random.gammavariate(0.5 (v_0 - D + 1), 2.0)`` (ONE shape for every row: the
  reference's parametrisation, not the textbook chi-square ladder);
* the factor is rounded to float32 before it is used.

``random`` and ``np.random`` are independent streams, so the normals are drawn in ONE vectorised call
(numpy's legacy Gaussian generator carries its spare value from call to call: one call of size n
equals the reference's D - 1 calls of sizes 1 .. D - 1 value for value) and scattered into the
triangle.  The products are associated as the reference associates them -- ``((C A) A') C'`` -- and the
inverse is ``solve(sample, I)``, so a draw equals the reference's bit for bit (tested with ``==``).
``rng`` / ``nprng`` replace the process-global generators (one chain per GPU).
"""
import random as _random

import numpy as np


def bartlett_factor(D, shape, rng=None, nprng=None):
    """Lower-triangular A (float32) with ``A A'`` ~ Wishart(I) in the reference's parametrisation."""
    rng = _random if rng is None else rng
    nprng = np.random if nprng is None else nprng
    A = np.zeros((D, D), dtype=np.float32)
    n_off = D * (D - 1) // 2
    if n_off:
        A[np.tril_indices(D, -1)] = nprng.normal(size=n_off)
    A[np.diag_indices(D)] = np.sqrt([rng.gammavariate(shape, 2.0) for _ in range(D)])
    return A


def wishrnd(sigma, v_0, C=None, rng=None, nprng=None):
    """One Wishart(sigma, v_0) sample; ``C`` is any factor of sigma (its Cholesky factor by default)."""
    sigma = np.asarray(sigma, dtype=float)
    D = sigma.shape[0]
    if C is None:
        C = np.linalg.cholesky(sigma)
    C = np.asarray(C, dtype=float)
    A = bartlett_factor(D, 0.5 * (v_0 - D + 1), rng, nprng).astype(float)
    return ((C @ A) @ A.T) @ C.T


def iwishrnd(sigma, v_0, C=None, rng=None, nprng=None):
    """One inverse-Wishart sample: the inverse of a Wishart(sigma, v_0) draw."""
    W = wishrnd(sigma, v_0, C, rng=rng, nprng=nprng)
    return np.linalg.solve(W, np.eye(W.shape[0]))
