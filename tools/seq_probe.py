"""Timing probe of the small-D sequential sweep: C2-like (movers dense) and a well-separated twin (hardly a move)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
for (N, D, K, sep) in ((100000, 2, 20, 1.0), (100000, 2, 20, 8.0), (100000, 4, 40, 3.0)):
    X, zt = gendata.synth_mixture(N, D, K, seed=1, mu_scale=sep)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    c = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 256)
    c.set_assignments(zt)
    rs = np.random.RandomState(0)
    for it in range(4):
        u = rs.random_sample(N)
        t0 = time.time(); c.sweep(u); dt = time.time() - t0
        st = c.sweep_stats()
    print("N=%d D=%d K=%d sep=%.1f: %.1f ms/sweep, %.3f us/visit, moves %d, K_end %d" % (
        N, D, K, sep, dt * 1e3, dt * 1e6 / N, st["moves"], c.K))
    c.close()
