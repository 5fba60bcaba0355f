"""Small mover-heavy run for profiling: rand init, one sweep."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X, zt = gendata.synth_mixture(N, D, K, seed=1)
m0, k0, v0, S0 = gendata.demo_prior_params(D)
rs = np.random.RandomState(3)
z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]
ctx = _lib.Context(X, m0, k0, v0, S0, 1.0, 4 * K)
ctx.set_assignments(z0)
for it in range(int(sys.argv[4]) if len(sys.argv) > 4 else 1):
    u = rs.random_sample(N)
    t = time.time(); ctx.sweep(u); dt = time.time() - t
    st = ctx.sweep_stats()
    print("sweep %d: %.3f s  %.2f us/visit  moves=%d steps=%d windows=%d K=%d  us/step=%.1f" % (
        it, dt, dt / N * 1e6, st["moves"], st["steps"], st["windows"], ctx.K, dt / max(st["steps"], 1) * 1e6))

print("path:", ctx.path_stats())
