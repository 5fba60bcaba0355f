"""Does a chain that starts far from equilibrium find its way back to the cheap kernels?  Per sweep:
time, moves, windows / steps and what the pruning / certification layers did.
    python tools/recovery_probe.py N D K [n_sweeps] [rand|flip] [full|diag|fixed] [prune_mode]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n_sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
init = sys.argv[5] if len(sys.argv) > 5 else "rand"
cov = sys.argv[6] if len(sys.argv) > 6 else "full"
prune_mode = int(sys.argv[7]) if len(sys.argv) > 7 else 0
X, zt = gendata.synth_mixture(N, D, K, seed=11)
m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
if cov == "diag":
    S_0 = np.ascontiguousarray(np.diag(S_0))
elif cov == "fixed":
    m_0, k_0, v_0 = np.zeros(D), 1.0, 1
    S_0 = np.concatenate([np.full(D, 0.49), np.full(D, 16.0)])
rs = np.random.RandomState(0)
if init == "rand":
    z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]
else:
    z0 = zt.copy()
    flip = rs.choice(N, size=max(N // 500, 1), replace=False)
    z0[flip] = rs.randint(0, K, size=flip.size)
ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, cov_type=cov)
ctx.set_tuning(prune_mode=prune_mode)
ctx.set_assignments(z0)
for it in range(n_sweeps):
    u = rs.random_sample(N)
    t0 = time.time(); ctx.sweep(u, None, None); dt = time.time() - t0
    st, ps = ctx.sweep_stats(), ctx.prune_stats()
    print("sweep %2d: %9.3f ms  moves %7d  windows %6d steps %6d  K %3d  certified %7d  bounded blocks %9d kept %8d" % (
        it, dt * 1e3, st["moves"], st["windows"], st["steps"], ctx.K, ps["certified_visits"], ps["bound_blocks"], ps["kept_blocks"]))
