import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N, D, K = 1000000, 16, 100
X, zt = gendata.synth_mixture(N, D, K, seed=1)
m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
rs = np.random.RandomState(0)
ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
ctx.set_assignments(zt)
for it in range(6):
    u = rs.random_sample(N)
    use_order = len(sys.argv) > 1
    order = rs.permutation(N) if use_order else None
    ctx.sweep(u, order, 1.01 if (it > 0 and use_order) else None)
    st = ctx.sweep_stats()
    print(it, "moves", st["moves"], "windows", st["windows"], ctx.prune_stats())
