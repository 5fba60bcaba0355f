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

import ctypes
out = (ctypes.c_int64 * 16)()
ctx.L.bgmm_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
ctx.L.bgmm_debug_prof(ctx.h, out)
v = list(out)
tot = (sum(v[:7]) + sum(v[8:12])) or 1
print("resolver ticks: setup %.1f%% A %.1f%% B %.1f%% C %.1f%% D1 %.1f%% D2 %.1f%% | total ticks %d calls %d ticks/call %.0f" % (
    *(100.0 * x / tot for x in v[:6]), tot, v[7], tot / max(v[7], 1)))

mv = max(st["moves"], 1)
print("ticks per mover: A %.0f | B: load+stats %.0f, p %.0f, scan %.0f, columns %.0f, outputs %.0f | C %.0f D1 %.0f pick %.0f" % (
    v[1] / mv, v[8] / mv, v[9] / mv, v[10] / mv, v[11] / mv, v[2] / mv, v[3] / mv, v[4] / mv, v[5] / mv))
