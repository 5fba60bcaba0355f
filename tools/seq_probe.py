import sys, os, time, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N, D, K = 100000, 2, 20
X, zt = gendata.synth_mixture(N, D, K, seed=1)
m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
c = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 256)
c.set_assignments(zt)
rs = np.random.RandomState(0)
for it in range(4):
    u = rs.random_sample(N)
    t0 = time.time(); c.sweep(u); dt = time.time() - t0
    st = c.sweep_stats()
    out = (ctypes.c_int64 * 16)()
    c.L.bgmm_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    c.L.bgmm_debug_prof(c.h, out)
    v = list(out)[:8]
    mv = max(st["moves"], 1)
    print("sweep %d: %.1f ms, moves %d K %d | ticks/visit: fetch %.0f find %.0f score %.0f | per mover: book %.0f stats %.0f rebuild %.0f wfrag %.0f | total ticks %d" % (
        it, dt * 1e3, st["moves"], c.K, v[0] / N, v[1] / N, v[2] / N, v[3] / mv, v[4] / mv, v[5] / mv, v[6] / mv, sum(v)))
