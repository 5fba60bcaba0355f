"""Development probe: average duration of the likelihood-kernel launches of one steady-state sweep
(chain initialised at the truth) at a given shape, through the library's HIP-event timing.
    python tools/prune_probe.py N D K [prune_mode]      (prune_mode 0 = pruning on, 1 = off)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
prune = int(sys.argv[4]) if len(sys.argv) > 4 else 0
X, zt = gendata.synth_mixture(N, D, K, seed=11)
m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
us = np.random.RandomState(0).random_sample((3, N))
ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
ctx.set_tuning(prune_mode=prune)
ctx.set_assignments(zt)
for it in range(2):
    ctx.sweep(us[it], None, None)
ctx.set_kernel_timing(True)
ctx.sweep(us[2], None, None)
n, ms = ctx.kernel_timing()
print("launches %d  avg %.4f ms" % (n, ms / max(n, 1)), ctx.sweep_stats(), ctx.prune_stats())
ctx.close()
