import sys, os, random
sys.path.insert(0, "/root/repo")
import numpy as np
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N = 1000000
X, zt = gendata.synth_mixture(N, 2, 3, seed=1)
ctx = _lib.Context(X, *gendata.demo_prior_params(2), 1.0, 12)
ctx.set_assignments(zt)
r = random.Random(5)
st = r.getstate()
key, pos = np.array(st[1][:624], dtype=np.uint32), int(st[1][624])
for it in range(24):
    key, pos = ctx.stage_mt19937(key, pos, None)
ctx.synchronize()
ctx.close()
print("done")
