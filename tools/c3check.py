"""C3-shaped check of home_kernel's neighbour path: the at-rest chain in the benchmarked mode against the full
evaluation, label for label, and how many visits are left to the pruning kernel."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sep = float(sys.argv[4]) if len(sys.argv) > 4 else 4.0
X, zt = gendata.synth_mixture(N, D, K, seed=3, mu_scale=sep)
m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
rs = np.random.RandomState(2)
ctxs = {}
for mode in (3, 1, 0):
    c = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
    c.set_tuning(prune_mode=mode)
    c.set_assignments(zt)
    ctxs[mode] = c
for it in range(4):
    u = rs.random_sample(N)
    order = rs.permutation(N).astype(np.int64)
    zs = {}
    for mode, c in ctxs.items():
        c.stage(u, order); c.synchronize()
        t0 = time.time(); c.sweep_staged(1.01 if it else None); c.synchronize(); dt = time.time() - t0
        zs[mode] = c.assignments()
        ps = c.path_stats()
        print("sweep %d mode %d: %.3f ms moves %d home-decided %d pairs %d" % (it, mode, dt * 1e3, c.sweep_stats()["moves"], ps["home_decided"], ps["pairs_executed"]))
    for mode in (3, 0):
        bad = np.nonzero(zs[mode] != zs[1])[0]
        print("   mode %d vs full evaluation: %s" % (mode, "identical" if bad.size == 0 else "%d labels DIFFER, first %d" % (bad.size, bad[0])))
