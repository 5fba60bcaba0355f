import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N = 1000000
X, zt = gendata.synth_mixture(N, 2, 3, seed=1)
ctx = _lib.Context(X, *gendata.demo_prior_params(2), 1.0, 12)
ctx.set_assignments(zt)
host = np.random.RandomState(1)
key, pos = host.get_state()[1].copy(), int(host.get_state()[2])
for it in range(6):
    key, pos = ctx.stage_permutation_mt19937(key, pos)
ts = []
for it in range(40):
    time.sleep(0.003)
    t0 = time.perf_counter()
    key, pos = ctx.stage_permutation_mt19937(key, pos)
    ts.append(time.perf_counter() - t0)
print("host time per call with the generations long finished: median %.1f us, min %.1f" % (np.median(ts) * 1e6, min(ts) * 1e6))
