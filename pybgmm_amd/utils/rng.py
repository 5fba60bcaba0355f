"""
Host-side consumption of the caller's seeded random streams.

The reference draws ONE ``random.random()`` per visited datapoint
(pybgmm/utils/utils.py:15) from the process-global Mersenne Twister, and the
pCRP sampler draws ``np.random.permutation(N)`` per sweep (pybgmm/igmm/pcrpmm.py:89).
To keep a user's ``random.seed(s); np.random.seed(s)`` meaning the same thing,
the sweep kernel takes those values as inputs and this module produces them from
the very same streams.

``take_uniforms`` is the vectorised, bit-identical route: CPython's ``random`` and
numpy's legacy ``RandomState`` share MT19937 and the ``genrand_res53`` double
formula, so the generator state is transplanted into a ``RandomState``, N doubles
are drawn at C speed, and the advanced state is written back.
"""
import random as _random

import numpy as np


def take_uniforms(n, rng=None):
    """``[rng.random() for _ in range(n)]`` as a float64 array, stream left exactly
    where n scalar calls would leave it.  ``rng``: the ``random`` module (default)
    or a ``random.Random`` instance (one per chain)."""
    rng = _random if rng is None else rng
    version, key, gauss_next = rng.getstate()
    if version != 3 or len(key) != 625:          # unknown layout: stay correct, be slow
        return np.array([rng.random() for _ in range(n)], dtype=np.float64)
    rs = np.random.RandomState()
    rs.set_state(("MT19937", np.asarray(key[:-1], dtype=np.uint32), int(key[-1])))
    u = rs.random_sample(int(n))
    _, new_key, pos = rs.get_state()[:3]
    rng.setstate((version, tuple(int(v) for v in new_key) + (int(pos),), gauss_next))
    return u


def take_permutation(n, nprng=None):
    """``np.random.permutation(range(n))`` from the global legacy stream (or a given
    ``RandomState``); identical to ``permutation(n)``."""
    nprng = np.random if nprng is None else nprng
    return np.asarray(nprng.permutation(int(n)), dtype=np.int64)
