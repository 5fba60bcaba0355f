"""Does a chain that starts far from equilibrium find its way back to the cheap kernels?  Per sweep:
time, moves, windows / steps and what the pruning / certification layers did.
    python tools/recovery_probe.py N D K [n_sweeps] [rand|flip]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n_sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
init = sys.argv[5] if len(sys.argv) > 5 else "rand"
X, zt = gendata.synth_mixture(N, D, K, seed=11)
m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
rs = np.random.RandomState(0)
if init == "rand":
    z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]
else:
    z0 = zt.copy()
    flip = rs.choice(N, size=max(N // 500, 1), replace=False)
    z0[flip] = rs.randint(0, K, size=flip.size)
ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
ctx.set_assignments(z0)
for it in range(n_sweeps):
    u = rs.random_sample(N)
    t0 = time.time(); ctx.sweep(u, None, None); dt = time.time() - t0
    st, ps = ctx.sweep_stats(), ctx.prune_stats()
    print("sweep %2d: %9.3f ms  moves %7d  windows %6d steps %6d  K %3d  certified %7d  bounded blocks %9d kept %8d" % (
        it, dt * 1e3, st["moves"], st["windows"], st["steps"], ctx.K, ps["certified_visits"], ps["bound_blocks"], ps["kept_blocks"]))
