"""Development probe: moves and certified visits per sweep, chain started from a perturbed truth."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nflip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
X, zt = gendata.synth_mixture(N, D, K, seed=11)
m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
rs = np.random.RandomState(0)
us = rs.random_sample((6, N))
z0 = zt.copy()
if nflip:
    flip = rs.choice(N, size=nflip, replace=False)
    z0[flip] = rs.randint(0, K, size=nflip)
ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
ctx.set_assignments(z0)
for it in range(6):
    ctx.sweep(us[it], None, None)
    st = ctx.sweep_stats()
    print(it, "moves", st["moves"], "windows", st["windows"], "steps", st["steps"], ctx.prune_stats())
