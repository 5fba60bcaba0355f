import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N, D, K = 200000, 64, 200
X, zt = gendata.synth_mixture(N, D, K, seed=11)
m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
us = np.random.RandomState(0).random_sample((4, N))
ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
ctx.set_assignments(zt)
for it in range(4):
    ctx.sweep(us[it], None, None)
    print(it, ctx.sweep_stats()["moves"], ctx.prune_stats())
