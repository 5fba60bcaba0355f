"""Development probe of home_kernel (kernels_home.hip): launch time of the pruned-window group of one
steady-state sweep with certified stays off, and -- when the library was built with -DBGMM_HOME_PROF
(tools/build_prof.sh) -- the per-phase shader clocks of its wavefronts.
    python tools/home_probe.py N D K [prof]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybgmm_amd import _build
if len(sys.argv) > 4 and sys.argv[4] == "prof":
    _build.LIB = os.path.join(os.path.dirname(_build.LIB), "libbgmm_hip_prof.so")
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X, zt = gendata.synth_mixture(N, D, K, seed=11)
m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
us = np.random.RandomState(0).random_sample((4, N))
ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
ctx.set_tuning(prune_mode=3)
ctx.set_assignments(zt)
for it in range(3):
    ctx.sweep(us[it], None, None)
pc0 = np.array(ctx.phase_clocks())
ctx.set_kernel_timing(True)
ctx.sweep(us[3], None, None)
n, ms = ctx.kernel_timing()
pc = np.array(ctx.phase_clocks()) - pc0
print("N=%d D=%d K=%d: launches %d  avg %.4f ms" % (N, D, K, n, ms / max(n, 1)), ctx.sweep_stats())
if pc[15] > 0:
    names = ["wait+stage", "issue", "frags", "mfma+reduce", "tail", "records", "blockhead", "switch"]
    tiles = N / 16.0
    tot = float(pc[:8].sum())
    for k in range(8):
        print("  %-12s %8.0f cycles per tile-wave  %5.1f %%" % (names[k], pc[k] / tiles, 100.0 * pc[k] / tot))
    print("  waves %d, total %.0f cycles per tile-wave" % (pc[15], tot / tiles))
ctx.close()
